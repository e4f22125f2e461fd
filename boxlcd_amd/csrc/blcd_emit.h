// blcd_emit.h — per-environment observation + LCD emission, shared by the fused rollout kernel (blcd_cfg.hip) and the
// stand-alone obs / render kernels (blcd_api.hip).
#pragma once
#include "blcd_raster.h"
#include "blcd_world.h"

namespace blcd {

constexpr int kBlock = 64;  // one wave per workgroup: N/64 workgroups spread over 256 CUs


// observation (boxLCD/world_env.py:387-429, float64 glue) + LCD raster (:460-512) of one environment; `body(i, &p, &a, &sel)`
// yields transform position, body angle and shape choice of body i.  Shared by obs_kernel and the fused rollout path.
template <int H, typename RowT, typename ObsT, bool CIRC = false, typename BodyFn>
__device__ __forceinline__ bool emit_env(const DevScene* __restrict__ S, BodyFn body, ObsT* __restrict__ obsRow,
                                         uint8_t* __restrict__ lcdRow, RowT* __restrict__ rowsOut = nullptr, bool lcdBits = false) {
  bool ok = true;
  if (obsRow) {
    int cachedKey = -1;   // (body, angle source) whose float64 sin/cos are in cs / cc: the cos and sin entries of a body are
    double cs = 0.0, cc = 1.0;   // adjacent in the obs table, so one sincos() serves both
    for (int i = 0; i < S->nobs; ++i) {
      const DevObs od = S->obs[i];
      Vec2 p;
      float a;
      int sel;
      body(od.body, &p, &a, &sel);
      double val;
      if (od.kind == 0) val = (double)p.x;
      else if (od.kind == 1) val = (double)p.y;
      else {
        const int key = od.body * 2 + (od.kind >= 4 ? 1 : 0);
        if (key != cachedKey) {
          if (od.kind >= 4) {  // transform.angle = atan2f(q.s, q.c)
            Rot q;
            q.Set(a);
            a = atan2f(q.s, q.c);
          }
          sincos((double)a, &cs, &cc);
          cachedKey = key;
        }
        val = (od.kind == 2 || od.kind == 4) ? cc : cs;
      }
      double lo = (double)od.lo, hi = (double)od.hi;
      obsRow[i] = (ObsT)(((val - lo) / (hi - lo) * 2.0) + -1.0);
    }
  }
  if (lcdRow || rowsOut) {
    Raster<H, RowT> r;
    r.clear(S->lcdW, S->rasterVariant);
    for (int i = 0; i < S->nb; ++i) {
      Vec2 p;
      float a;
      int sel;
      body(i, &p, &a, &sel);
      Transform xf;
      xf.p = p;
      if (CIRC) {  // a circle's frame does not depend on its rotation
        xf.q.s = 0.0f;
        xf.q.c = 1.0f;
      } else {
        xf.q.Set(a);
      }
      ok = r.template drawBody<CIRC>(&S->shapes[S->bodies[i].var[sel].shape], xf, (double)S->worldW, (double)S->lcdW) && ok;
    }
    if (rowsOut) {   // the caller writes the frame itself (wave-coalesced, see step_kernel): rows in output order
#pragma unroll
      for (int y = 0; y < H; ++y) rowsOut[y] = r.rows[H - 1 - y];
    } else if (lcdBits) {
      r.writeBits(lcdRow);
    } else {
      r.write(lcdRow);
    }
  }
  return ok;
}

}  // namespace blcd
