// blcd_render_ex.h — `lcd_render(width, height, lcd_mode)` on the device for any canvas size and for mode 'RGB'
// (boxLCD/world_env.py:460-512; the 8x human view of render(), :514-535).  One wave renders one frame, lane = image row:
// rows are independent in Pillow's scan conversion, so every lane replays, for ITS row only, what Pillow does body by body:
// polygon fill (polygon_generic's spans of that scanline + the horizontal-edge spans), then the outline's Bresenham lines
// (ImagingDrawLine from each vertex to the next, direction-dependent), ellipse fill/outline from Pillow's span table.
// The test-side checker draws whole canvases sequentially like Pillow; both are checked against Pillow goldens.
#pragma once
#include "blcd_world.h"

namespace blcd {

struct RowCanvas {
  uint8_t* row;   // this lane's output row (already flipped), W * C bytes
  int W, C, y, H;
  bool inv;       // RGB mode stores 255 - ink (world_env.py:510-511)
  __device__ void span(int x0, int x1, const uint8_t* ink) const {
    if (x0 < 0) x0 = 0;
    else if (x0 >= W) return;
    if (x1 < 0) return;
    else if (x1 >= W) x1 = W - 1;
    for (int x = x0; x <= x1; ++x)
      for (int c = 0; c < C; ++c) row[x * C + c] = inv ? (uint8_t)(255 - ink[c]) : ink[c];
  }
};

__device__ inline int rexRoundUp(float f) { return (int)(f >= 0.0f ? floor((double)(f + 0.5f)) : -floor((double)(fabsf(f) + 0.5f))); }
__device__ inline int rexRoundDown(float f) { return (int)(f >= 0.0f ? ceil((double)(f - 0.5f)) : -ceil((double)(fabsf(f) - 0.5f))); }

// scanline `cv.y` of ImagingDrawPolygon(fill=1) on integer vertices
__device__ inline void rexPolygonFillRow(const RowCanvas& cv, const int* xy, int count, int variant, const uint8_t* ink) {
  struct E { int xmin, ymin, xmax, ymax, x0, y0; float dx; };
  E e[kShapeVerts + 1];
  int n = 0;
  auto add = [&](int x0, int y0, int x1, int y1) {
    E& q = e[n++];
    q.xmin = x0 <= x1 ? x0 : x1; q.xmax = x0 <= x1 ? x1 : x0;
    q.ymin = y0 <= y1 ? y0 : y1; q.ymax = y0 <= y1 ? y1 : y0;
    q.dx = y0 == y1 ? 0.0f : ((float)(x1 - x0)) / (float)(y1 - y0);
    q.x0 = x0; q.y0 = y0;
  };
  for (int i = 0; i < count - 1; ++i) add(xy[2 * i], xy[2 * i + 1], xy[2 * i + 2], xy[2 * i + 3]);
  if (xy[2 * (count - 1)] != xy[0] || xy[2 * (count - 1) + 1] != xy[1]) add(xy[2 * (count - 1)], xy[2 * (count - 1) + 1], xy[0], xy[1]);
  const int y = cv.y, H = cv.H;
  int ymin = H - 1, ymax = 0;
  int table[kShapeVerts + 1], ne = 0;
  for (int i = 0; i < n; ++i) {
    if (ymin > e[i].ymin) ymin = e[i].ymin;
    if (ymax < e[i].ymax) ymax = e[i].ymax;
    if (e[i].ymin == e[i].ymax) {
      if (variant != 2 && e[i].ymin == y) cv.span(e[i].xmin, e[i].xmax, ink);
      continue;
    }
    table[ne++] = i;
  }
  if (ymin < 0) ymin = 0;
  if (ymax > H) ymax = H;
  if (y < ymin || y > ymax) return;
  auto X = [&](const E& q, int yy) { return (float)(yy - q.y0) * q.dx + (float)q.x0; };
  float xx[2 * (kShapeVerts + 1)];
  int j = 0;
  for (int i = 0; i < ne; ++i) {
    const E& cur = e[table[i]];
    if (y >= cur.ymin && y <= cur.ymax) {
      xx[j++] = X(cur, y);
      if (y == cur.ymax && y < ymax) {
        xx[j] = xx[j - 1];
        j++;
      } else if (variant == 1 && cur.dx != 0.0f && roundf(xx[j - 1]) == xx[j - 1]) {
        for (int k = 0; k < i; ++k) {
          const E& oth = e[table[k]];
          if ((cur.dx > 0 && oth.dx <= 0) || (cur.dx < 0 && oth.dx >= 0)) continue;
          if (!((y == cur.ymin && y == oth.ymin) || (y == cur.ymax && y == oth.ymax))) continue;
          if (xx[j - 1] == X(oth, y)) {
            const int off = (y == ymax) ? -1 : 1;
            const float a = X(cur, y + off), b = X(oth, y + off);
            int v;
            if (y == cur.ymax) v = cur.dx > 0 ? rexRoundUp(a > b ? a : b) + 1 : rexRoundUp(a < b ? a : b) - 1;
            else v = cur.dx > 0 ? rexRoundUp(a < b ? a : b) - 1 : rexRoundUp(a > b ? a : b) + 1;
            const bool want_left = (y == cur.ymax) ? (cur.dx > 0) : (cur.dx < 0);
            const float corner_x = xx[j - 1];
            if ((float)v == corner_x || (((float)v < corner_x) == want_left)) xx[j - 1] = (float)v;
            break;
          }
        }
      }
    }
  }
  for (int p = 1; p < j; ++p) {
    float key = xx[p];
    int t = p - 1;
    while (t >= 0 && xx[t] > key) { xx[t + 1] = xx[t]; --t; }
    xx[t + 1] = key;
  }
  if (variant == 2) {   // as Raster::polygon (blcd_raster.h): scan position from 0, horizontal edges from that position only
    int x_pos = 0;
    auto horizontalLines = [&]() {
      for (int k = 0; k < n; ++k) {
        if (e[k].ymin != y || e[k].ymin != e[k].ymax) continue;
        int xmin = e[k].xmin;
        if (x_pos < xmin) continue;
        const int xmax = e[k].xmax;
        if (x_pos > xmin) {
          xmin = x_pos;
          if (xmax < xmin) continue;
        }
        cv.span(xmin, xmax, ink);
        x_pos = xmax + 1;
      }
    };
    for (int i = 1; i < j; i += 2) {
      const int x_end = rexRoundDown(xx[i]);
      if (x_end < x_pos) continue;
      horizontalLines();
      if (x_end < x_pos) continue;
      int x_start = rexRoundUp(xx[i - 1]);
      if (x_pos > x_start) {
        x_start = x_pos;
        if (x_end < x_start) continue;
      }
      cv.span(x_start < x_end ? x_start : x_end, x_start < x_end ? x_end : x_start, ink);
      x_pos = x_end + 1;
    }
    horizontalLines();
    return;
  }
  int x_pos = 0;
  for (int i = 1; i < j; i += 2) {
    const int x_end = rexRoundDown(xx[i]);
    if (x_end < x_pos) continue;
    int x_start = rexRoundUp(xx[i - 1]);
    if (x_pos > x_start) {
      x_start = x_pos;
      if (x_end < x_start) continue;
    }
    cv.span(x_start, x_end, ink);
    x_pos = x_end + 1;
  }
}

// the pixels of one ImagingDrawLine (Bresenham, both ends, error term seeded at the first point) that fall on row cv.y
__device__ inline void rexLineRow(const RowCanvas& cv, int x0, int y0, int x1, int y1, const uint8_t* ink) {
  int dx = x1 - x0, dy = y1 - y0, xs = 1, ys = 1;
  if (dx < 0) { dx = -dx; xs = -1; }
  if (dy < 0) { dy = -dy; ys = -1; }
  const int y = cv.y;
  if ((y0 < y1 ? y0 : y1) > y || (y0 < y1 ? y1 : y0) < y) return;
  if (dx == 0) {
    cv.span(x0, x0, ink);
  } else if (dy == 0) {
    cv.span(x0 < x1 ? x0 : x1, x0 < x1 ? x1 : x0, ink);
  } else if (dx > dy) {
    const int n = dx;
    dy += dy;
    int e = dy - dx;
    dx += dx;
    for (int i = 0; i <= n; ++i) {
      if (y0 == y) cv.span(x0, x0, ink);
      if (e >= 0) { y0 += ys; e -= dx; }
      e += dy;
      x0 += xs;
    }
  } else {
    const int n = dy;
    dx += dx;
    int e = dx - dy;
    dy += dy;
    for (int i = 0; i <= n; ++i) {
      if (y0 == y) cv.span(x0, x0, ink);
      if (e >= 0) { x0 += xs; e -= dy; }
      e += dx;
      y0 += ys;
    }
  }
}

// frame k: poses [nb][3], one lane per image row; lut = Pillow's ellipse span table uint8 [amax+1][5][amax+3][6]
__global__ void render_ex_kernel(const DevScene* __restrict__ S, int m, const float* __restrict__ poses, const int* __restrict__ shapeSel,
                                 int W, int H, int mode, const uint8_t* __restrict__ lut, int amax, uint8_t* __restrict__ out,
                                 int* __restrict__ err) {
  const int k = blockIdx.x;
  if (k >= m) return;
  const int C = mode ? 3 : 1;
  const int nb = S->nb;
  const double WIDTH = (double)S->worldW, width = (double)W;
  for (int y = threadIdx.x; y < H; y += blockDim.x) {
    RowCanvas cv;
    cv.row = out + ((size_t)k * H + (size_t)(H - 1 - y)) * W * C;   // FLIP_TOP_BOTTOM
    cv.W = W; cv.C = C; cv.y = y; cv.H = H; cv.inv = mode != 0;
    const uint8_t bg[3] = {1, 1, 1};
    cv.span(0, W - 1, bg);
    for (int i = 0; i < nb; ++i) {
      const float* p = poses + ((size_t)k * nb + i) * 3;
      Transform xf;
      xf.p = V2(p[0], p[1]);
      xf.q.Set(p[2]);
      int sel = shapeSel ? shapeSel[(size_t)k * nb + i] : 0;
      if (sel < 0 || sel >= S->bodies[i].nChoices) sel = 0;
      const Shape* sh = &S->shapes[S->bodies[i].var[sel].shape];
      const bool robot = S->bodies[i].nJoints > 0 || S->bodyKind[i] != 0;
      // int(255.0 * (1 - x)) of body.color1 / color2 (world_env.py:201,303,482-483); mode '1': fill 0, no outline
      const uint8_t fillc[3] = {(uint8_t)(mode ? (robot ? 25 : 127) : 0), (uint8_t)(mode ? 153 : 0), (uint8_t)(mode ? (robot ? 153 : 25) : 0)};
      const uint8_t outc[3] = {(uint8_t)(robot ? 127 : 178), 178, 127};
      if (sh->type == kCircle) {
        const double px = (double)xf.p.x, py = (double)xf.p.y, rad = (double)sh->radius;
        const int x0 = (int)((px - rad) / WIDTH * width), y0 = (int)((py - rad) / WIDTH * width);
        const int x1 = (int)((px + rad) / WIDTH * width), y1 = (int)((py + rad) / WIDTH * width);
        const int a = x1 - x0, b = y1 - y0;
        if (a < 0 || b < 0 || a > amax || b - a < -2 || b - a > 2) {
          if (err) *err = 1;
          continue;
        }
        const int r = y - y0;
        if (r < 0 || r > b) continue;
        const uint8_t* t = lut + (((size_t)(a * 5 + (b - a + 2)) * (amax + 3)) + r) * 6;
        if (t[0] != 255) cv.span(x0 + t[0], x0 + t[1], fillc);
        if (mode) {
          if (t[2] != 255) cv.span(x0 + t[2], x0 + t[3], outc);
          if (t[4] != 255) cv.span(x0 + t[4], x0 + t[5], outc);
        }
      } else {
        int xy[2 * kShapeVerts];
        for (int q = 0; q < sh->count; ++q) {
          const Vec2 v = Mul(xf, sh->v[q]);
          xy[2 * q] = (int)((double)v.x / WIDTH * width);
          xy[2 * q + 1] = (int)((double)v.y / WIDTH * width);
        }
        rexPolygonFillRow(cv, xy, sh->count, S->rasterVariant, fillc);
        if (mode) {
          for (int q = 0; q < sh->count - 1; ++q) rexLineRow(cv, xy[2 * q], xy[2 * q + 1], xy[2 * q + 2], xy[2 * q + 3], outc);
          rexLineRow(cv, xy[2 * (sh->count - 1)], xy[2 * (sh->count - 1) + 1], xy[0], xy[1], outc);
        }
      }
    }
  }
}

}  // namespace blcd
