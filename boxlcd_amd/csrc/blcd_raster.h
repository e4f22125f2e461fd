// blcd_raster.h — on-device 1-bit LCD rasteriser: one thread renders one environment into H row bit-masks.
//
// Replaces `WorldEnv.lcd_render` (boxLCD/world_env.py:460-512): Image.new('1') + draw.rectangle(bg=1) +
// draw.ellipse / draw.polygon(fill=0) per dynamic body + FLIP_TOP_BOTTOM.  The scan-conversion rules are Pillow's
// (`ImagingDrawPolygon` / `polygon_generic` / `hline8`, `ImagingDrawEllipse`) as characterised in SURVEY.md App. C:
// int truncation of the float64 pixel coordinates, float32 edge interpolation, half-open RU/RD rounding, lower-end
// duplication; variant 0 is the rule of the Pillow the reference pins (9.0.1: horizontal-edge spans + span merging),
// variant 1 adds Pillow >= 12's sub-pixel corner joining, variant 2 is the rule the reference's published demo GIFs were
// recorded with (each pixel once from a scan position that starts at 0, horizontal edges drawn from that position only,
// crossed span ends swapped: the behaviour of the Pillow 9.0.x the reference pins): they differ only on thin links that
// truncate to degenerate polygons.  Drawing order is irrelevant in mode '1' (every body clears bits).
#pragma once
#include "blcd_collide.h"
#include "blcd_ellipse_lut.h"

namespace blcd {

constexpr int kEllipseAmax = KBLCDELLIPSELUT_AMAX;

template <int H, typename RowT>
struct Raster {
  RowT rows[H];  // bit x of rows[y] set => body pixel at image (x, y), y down, before the vertical flip
  int W;
  int variant;

  __device__ void clear(int w, int var) {
    W = w;
    variant = var;
    for (int y = 0; y < H; ++y) rows[y] = 0;
  }
  __device__ void hline(int x0, int y, int x1) {
    if (y >= 0 && y < H) {
      if (x0 < 0) x0 = 0;
      else if (x0 >= W) return;
      if (x1 < 0) return;
      else if (x1 >= W) x1 = W - 1;
      if (x0 <= x1) {
        int n = x1 - x0 + 1;
        RowT m = (n >= (int)(8 * sizeof(RowT))) ? ~(RowT)0 : ((((RowT)1) << n) - 1);
        rows[y] |= m << x0;
      }
    }
  }
  static __device__ int RoundUp(float f) { return (int)(f >= 0.0f ? floor((double)(f + 0.5f)) : -floor((double)(fabsf(f) + 0.5f))); }
  static __device__ int RoundDown(float f) { return (int)(f >= 0.0f ? ceil((double)(f - 0.5f)) : -ceil((double)(fabsf(f) - 0.5f))); }

  struct Edge {
    int xmin, ymin, xmax, ymax, x0, y0;
    float dx;
  };
  static __device__ void addEdge(Edge* e, int x0, int y0, int x1, int y1) {
    if (x0 <= x1) { e->xmin = x0; e->xmax = x1; } else { e->xmin = x1; e->xmax = x0; }
    if (y0 <= y1) { e->ymin = y0; e->ymax = y1; } else { e->ymin = y1; e->ymax = y0; }
    if (y0 == y1) e->dx = 0.0f; else e->dx = ((float)(x1 - x0)) / (float)(y1 - y0);
    e->x0 = x0;
    e->y0 = y0;
  }
  static __device__ float edgeX(const Edge& e, int y) { return (float)(y - e.y0) * e.dx + (float)e.x0; }

  // filled polygon on integer vertices xy[2*count]
  __device__ void polygon(const int* xy, int count) {
    Edge e[kShapeVerts + 1];
    int n = 0;
    for (int i = 0; i < count - 1; i++) addEdge(&e[n++], xy[i * 2], xy[i * 2 + 1], xy[i * 2 + 2], xy[i * 2 + 3]);
    if (xy[(count - 1) * 2] != xy[0] || xy[(count - 1) * 2 + 1] != xy[1])
      addEdge(&e[n++], xy[(count - 1) * 2], xy[(count - 1) * 2 + 1], xy[0], xy[1]);
    int table[kShapeVerts + 1];
    int edge_count = 0;
    int ymin = H - 1, ymax = 0;
    for (int i = 0; i < n; i++) {
      if (ymin > e[i].ymin) ymin = e[i].ymin;
      if (ymax < e[i].ymax) ymax = e[i].ymax;
      if (e[i].ymin == e[i].ymax) {
        if (variant != 2) hline(e[i].xmin, e[i].ymin, e[i].xmax);
        continue;
      }
      table[edge_count++] = i;
    }
    if (ymin < 0) ymin = 0;
    if (ymax > H) ymax = H;
    float xx[2 * (kShapeVerts + 1)];
    for (int y = ymin; y <= ymax; y++) {
      int j = 0;
      for (int i = 0; i < edge_count; i++) {
        const Edge& cur = e[table[i]];
        if (y >= cur.ymin && y <= cur.ymax) {
          xx[j++] = edgeX(cur, y);
          if (y == cur.ymax && y < ymax) {
            xx[j] = xx[j - 1];
            j++;
          } else if (variant == 1 && cur.dx != 0.0f && roundf(xx[j - 1]) == xx[j - 1]) {
            for (int k = 0; k < i; k++) {
              const Edge& oth = e[table[k]];
              if ((cur.dx > 0 && oth.dx <= 0) || (cur.dx < 0 && oth.dx >= 0)) continue;
              if (!((y == cur.ymin && y == oth.ymin) || (y == cur.ymax && y == oth.ymax))) continue;
              if (xx[j - 1] == edgeX(oth, y)) {
                int off = (y == ymax) ? -1 : 1;
                float a = edgeX(cur, y + off), b = edgeX(oth, y + off);
                int v;
                if (y == cur.ymax) v = cur.dx > 0 ? RoundUp(a > b ? a : b) + 1 : RoundUp(a < b ? a : b) - 1;
                else v = cur.dx > 0 ? RoundUp(a < b ? a : b) - 1 : RoundUp(a > b ? a : b) + 1;
                bool want_left = (y == cur.ymax) ? (cur.dx > 0) : (cur.dx < 0);
                float corner_x = xx[j - 1];
                if ((float)v == corner_x || (((float)v < corner_x) == want_left)) xx[j - 1] = (float)v;
                break;
              }
            }
          }
        }
      }
      for (int p = 1; p < j; ++p) {  // insertion sort (j <= 2*(verts+1))
        float key = xx[p];
        int t = p - 1;
        while (t >= 0 && xx[t] > key) {
          xx[t + 1] = xx[t];
          --t;
        }
        xx[t + 1] = key;
      }
      if (variant == 2) {
        // each pixel once, from a scan position that starts at 0; this row's horizontal edges are drawn from that position, and one
        // that begins to the right of it is skipped (the one-pixel-high polygons of the recordings: DESIGN.md 2.1)
        int x_pos = 0;
        auto horizontalLines = [&]() {
          for (int k = 0; k < n; k++) {
            if (e[k].ymin != y || e[k].ymin != e[k].ymax) continue;
            int xmin = e[k].xmin;
            if (x_pos < xmin) continue;
            const int xmax = e[k].xmax;
            if (x_pos > xmin) {
              xmin = x_pos;
              if (xmax < xmin) continue;
            }
            hline(xmin, y, xmax);
            x_pos = xmax + 1;
          }
        };
        for (int i = 1; i < j; i += 2) {
          const int x_end = RoundDown(xx[i]);
          if (x_end < x_pos) continue;
          horizontalLines();
          if (x_end < x_pos) continue;
          int x_start = RoundUp(xx[i - 1]);
          if (x_pos > x_start) {
            x_start = x_pos;
            if (x_end < x_start) continue;
          }
          hline(x_start < x_end ? x_start : x_end, y, x_start < x_end ? x_end : x_start);   // crossed ends are swapped
          x_pos = x_end + 1;
        }
        horizontalLines();
        continue;
      }
      int x_pos = 0;
      for (int i = 1; i < j; i += 2) {
        int x_end = RoundDown(xx[i]);
        if (x_end < x_pos) continue;
        int x_start = RoundUp(xx[i - 1]);
        if (x_pos > x_start) {
          x_start = x_pos;
          if (x_end < x_start) continue;
        }
        hline(x_start, y, x_end);
        x_pos = x_end + 1;
      }
    }
  }

  // filled ellipse on the truncated bbox, Pillow span table
  __device__ bool ellipse(int x0, int y0, int x1, int y1) {
    int a = x1 - x0, b = y1 - y0;
    if (a < 0 || b < 0 || a > kEllipseAmax || b > kEllipseAmax) return false;
    for (int r = 0; r <= b; r++) {
      int s = kBlcdEllipseLut_data[a][b][r][0], t = kBlcdEllipseLut_data[a][b][r][1];
      if (s > t) continue;
      hline(x0 + s, y0 + r, x0 + t);
    }
    return true;
  }

  // one body: boxLCD/world_env.py:493-505 (float64 `/ WIDTH * width`, then Pillow's (int) truncation)
  template <bool CIRC = false>
  __device__ bool drawBody(const Shape* sh, const Transform& xf, double WIDTH, double width) {
    if (CIRC || sh->type == kCircle) {
      double px = (double)xf.p.x, py = (double)xf.p.y, rad = (double)sh->radius;
      double tlx = (px - rad) / WIDTH * width, tly = (py - rad) / WIDTH * width;
      double brx = (px + rad) / WIDTH * width, bry = (py + rad) / WIDTH * width;
      return ellipse((int)tlx, (int)tly, (int)brx, (int)bry);
    }
    int xy[2 * kShapeVerts];
    for (int k = 0; k < sh->count; ++k) {
      Vec2 p = Mul(xf, sh->v[k]);
      xy[2 * k] = (int)((double)p.x / WIDTH * width);
      xy[2 * k + 1] = (int)((double)p.y / WIDTH * width);
    }
    polygon(xy, sh->count);
    return true;
  }

  // the same frame at ONE BIT per pixel: uint8 [H][W / 8], pixel x of a row = bit (x % 8) of byte x / 8 (numpy bitorder='little'),
  // 1 = background, 0 = body, rows flipped.  This is what the raster holds anyway (the complement of its row masks).
  __device__ void writeBits(uint8_t* __restrict__ out) const {
    for (int r = 0; r < H; ++r) {
      const RowT m = ~rows[H - 1 - r];
      uint8_t* o = out + r * (W / 8);
      for (int x0 = 0; x0 < W; x0 += 8) o[x0 / 8] = (uint8_t)((m >> x0) & 0xffu);
    }
  }

  // uint8 [H][W], 1 = background, 0 = body, rows flipped (FLIP_TOP_BOTTOM)
  __device__ void write(uint8_t* __restrict__ out) const {
    for (int r = 0; r < H; ++r) {
      RowT m = rows[H - 1 - r];
      uint8_t* o = out + r * W;
      for (int x0 = 0; x0 < W; x0 += 8) {  // W is a multiple of 8 for every boxLCD env (16, 24, 32, 64)
        // 4 mask bits -> 4 bytes: x * (1 + 2^7 + 2^14 + 2^21) puts bit j at position 8j (no carries: the four shifted copies
        // of a 4-bit value do not overlap); body bit set -> pixel 0, background -> 1
        const uint32_t b8 = (uint32_t)(m >> x0) & 0xffu;
        const uint32_t lo = (((b8 & 0xfu) * 0x00204081u) & 0x01010101u) ^ 0x01010101u;
        const uint32_t hi = (((b8 >> 4) * 0x00204081u) & 0x01010101u) ^ 0x01010101u;
        *reinterpret_cast<uint2*>(o + x0) = make_uint2(lo, hi);
      }
    }
  }
};

}  // namespace blcd
