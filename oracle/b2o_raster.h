// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under boxlcd_amd/ may include, link or call this.
//
// CPU restatement of the Pillow calls made by boxLCD's `WorldEnv.lcd_render` in mode '1'
// (reference: boxLCD/world_env.py:460-512: Image.new / draw.rectangle(bg=1) / draw.ellipse / draw.polygon /
// FLIP_TOP_BOTTOM).  Pillow (`Pillow==9.0.1`, requirements.txt:19) is un-vendored; the scan-conversion rules are
// the behavioural spec of SURVEY.md App. C (probed against Pillow 12.2.0):
//   variant 1 "modern"  = Pillow 12.2.0 (with sub-pixel corner joining)      -> pinned by tests/golden/pillow_*.npz
//   variant 0 "legacy"  = variant 1's scan rule without corner joining (no fixture of its own)
//   variant 2 "recording era" = what the reference's published GIFs were rendered with (Pillow 9.0.x by its behaviour, see below)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include "b2o_ellipse_lut.h"

namespace b2o {

struct Canvas {
  int W, H;
  uint8_t* px;  // H*W*C; mode '1': C = 1, 1 = background, 0 = body; image coordinates (y down), not yet flipped
  int C = 1;
  uint8_t ink[3] = {0, 0, 0};   // current drawing colour
};

static inline void hline(Canvas& im, int x0, int y0, int x1) {
  if (y0 >= 0 && y0 < im.H) {
    if (x0 < 0) x0 = 0;
    else if (x0 >= im.W) return;
    if (x1 < 0) return;
    else if (x1 >= im.W) x1 = im.W - 1;
    for (int x = x0; x <= x1; ++x)
      for (int c = 0; c < im.C; ++c) im.px[((size_t)y0 * im.W + x) * im.C + c] = im.ink[c];
  }
}
static inline void point(Canvas& im, int x, int y) {
  if (x >= 0 && x < im.W && y >= 0 && y < im.H)
    for (int c = 0; c < im.C; ++c) im.px[((size_t)y * im.W + x) * im.C + c] = im.ink[c];
}

// ImagingDrawLine (Draw.c line8/line32): Bresenham from (x0,y0) to (x1,y1), both end points drawn; the error term starts from
// the FIRST point, so the pixel pattern depends on the direction (probed against Pillow 12.2, tests/test_oracle_raster.py)
static inline void draw_line(Canvas& im, int x0, int y0, int x1, int y1) {
  int dx = x1 - x0, dy = y1 - y0, xs = 1, ys = 1;
  if (dx < 0) dx = -dx, xs = -1;
  if (dy < 0) dy = -dy, ys = -1;
  if (dx == 0) {
    for (int i = 0; i <= dy; ++i, y0 += ys) point(im, x0, y0);
  } else if (dy == 0) {
    for (int i = 0; i <= dx; ++i, x0 += xs) point(im, x0, y0);
  } else if (dx > dy) {
    int n = dx;
    dy += dy;
    int e = dy - dx;
    dx += dx;
    for (int i = 0; i <= n; ++i) {
      point(im, x0, y0);
      if (e >= 0) { y0 += ys; e -= dx; }
      e += dy;
      x0 += xs;
    }
  } else {
    int n = dy;
    dx += dx;
    int e = dx - dy;
    dy += dy;
    for (int i = 0; i <= n; ++i) {
      point(im, x0, y0);
      if (e >= 0) { x0 += xs; e -= dy; }
      e += dx;
      y0 += ys;
    }
  }
}
// ImagingDrawPolygon(fill=0, width=1): the edges in order, then the closing edge
static inline void draw_polygon_outline(Canvas& im, const int* xy, int count) {
  for (int i = 0; i < count - 1; ++i) draw_line(im, xy[2 * i], xy[2 * i + 1], xy[2 * i + 2], xy[2 * i + 3]);
  draw_line(im, xy[2 * (count - 1)], xy[2 * (count - 1) + 1], xy[0], xy[1]);
}

// Ellipse with fill + 1-px outline on a near-round bbox: Pillow's span table (tools/gen_ellipse_rgb_lut.py), handed over by
// the loader (pyb2o.load -> b2o_set_ellipse_rgb_lut).  uint8 [amax+1][5][amax+3][6]
static const uint8_t* g_ellipse_rgb_lut = nullptr;
static int g_ellipse_rgb_amax = -1;
static inline bool draw_ellipse_rgb(Canvas& im, int x0, int y0, int x1, int y1, const uint8_t* fill, const uint8_t* outline) {
  int a = x1 - x0, b = y1 - y0;
  if (a < 0 || b < 0 || a > g_ellipse_rgb_amax || b - a < -2 || b - a > 2 || !g_ellipse_rgb_lut) return false;
  const uint8_t* t = g_ellipse_rgb_lut + ((size_t)(a * 5 + (b - a + 2)) * (g_ellipse_rgb_amax + 3)) * 6;
  if (fill) {
    std::memcpy(im.ink, fill, 3);
    for (int r = 0; r <= b; ++r)
      if (t[6 * r] != 255) hline(im, x0 + t[6 * r], y0 + r, x0 + t[6 * r + 1]);
  }
  if (outline) {
    std::memcpy(im.ink, outline, 3);
    for (int r = 0; r <= b; ++r)
      for (int q = 2; q < 6; q += 2)
        if (t[6 * r + q] != 255) hline(im, x0 + t[6 * r + q], y0 + r, x0 + t[6 * r + q + 1]);
  }
  return true;
}

static inline int RoundUp(float f) { return (int)(f >= 0.0f ? floor(f + 0.5f) : -floor(fabs(f) + 0.5f)); }
static inline int RoundDown(float f) { return (int)(f >= 0.0f ? ceil(f - 0.5f) : -ceil(fabs(f) - 0.5f)); }

struct Edge {
  int xmin, ymin, xmax, ymax;
  int x0, y0;
  float dx;
};
static inline void add_edge(Edge* e, int x0, int y0, int x1, int y1) {
  if (x0 <= x1) e->xmin = x0, e->xmax = x1; else e->xmin = x1, e->xmax = x0;
  if (y0 <= y1) e->ymin = y0, e->ymax = y1; else e->ymin = y1, e->ymax = y0;
  if (y0 == y1) e->dx = 0.0f; else e->dx = ((float)(x1 - x0)) / (y1 - y0);
  e->x0 = x0;
  e->y0 = y0;
}
static inline float edge_x(const Edge* e, int y) { return (y - e->y0) * e->dx + e->x0; }

// Variant 2, "recording era" = the polygon fill the reference's demo GIFs were rendered with, and by every sign the release the
// reference pins (requirements.txt: Pillow==9.0.1): the first release that draws each polygon pixel once (a scan position per row;
// horizontal edges are not drawn up front but from that position) and the last before the one-pixel-high-polygon fix - a horizontal
// edge that begins to the right of the scan position is skipped, and the position starts at 0.  Evidence (tools/degenerate_polys.py,
// profiles/r04_param_sweep.md): of the eleven robot links of the exactly replayed recording frames that truncate to a single row, the
// recordings draw exactly the two whose row begins at x = 0 (Luxo frame 37, LuxoBall frame 38) and none of the other nine; this
// rule reproduces all eleven and changes no other pixel of the 1 126 recorded frames.  Relative to variant 0 also: a span whose
// rounded ends cross (two coincident crossings at k + 0.5) is drawn with its ends swapped, as that era's hline did.
// The product implements the same rule (blcd_raster.h, blcd_render_ex.h) and uses it by default.
static inline void hline_swapping(Canvas& im, int x0, int y0, int x1) {
  if (x0 > x1) std::swap(x0, x1);
  hline(im, x0, y0, x1);
}

// ImagingDrawPolygon(fill) on integer vertices
static inline void draw_polygon(Canvas& im, const int* xy, int count, int variant) {
  const int MAXE = 40;
  if (count <= 0 || count + 1 > MAXE) return;
  Edge e[MAXE];
  int n = 0;
  for (int i = 0; i < count - 1; i++) add_edge(&e[n++], xy[i * 2], xy[i * 2 + 1], xy[i * 2 + 2], xy[i * 2 + 3]);
  if (xy[(count - 1) * 2] != xy[0] || xy[(count - 1) * 2 + 1] != xy[1])
    add_edge(&e[n++], xy[(count - 1) * 2], xy[(count - 1) * 2 + 1], xy[0], xy[1]);
  const Edge* table[MAXE];
  int edge_count = 0;
  int ymin = im.H - 1, ymax = 0;
  for (int i = 0; i < n; i++) {
    if (ymin > e[i].ymin) ymin = e[i].ymin;
    if (ymax < e[i].ymax) ymax = e[i].ymax;
    if (e[i].ymin == e[i].ymax) {
      if (variant != 2) hline(im, e[i].xmin, e[i].ymin, e[i].xmax);
      continue;
    }
    table[edge_count++] = &e[i];
  }
  if (ymin < 0) ymin = 0;
  if (ymax > im.H) ymax = im.H;
  float xx[2 * MAXE];
  for (int y = ymin; y <= ymax; y++) {
    int j = 0;
    for (int i = 0; i < edge_count; i++) {
      const Edge* cur = table[i];
      if (y >= cur->ymin && y <= cur->ymax) {
        xx[j++] = edge_x(cur, y);
        if (y == cur->ymax && y < ymax) {
          xx[j] = xx[j - 1];
          j++;
        } else if (variant == 1 && cur->dx != 0.0f && roundf(xx[j - 1]) == xx[j - 1]) {
          for (int k = 0; k < i; k++) {
            const Edge* oth = table[k];
            if ((cur->dx > 0 && oth->dx <= 0) || (cur->dx < 0 && oth->dx >= 0)) continue;
            if (!((y == cur->ymin && y == oth->ymin) || (y == cur->ymax && y == oth->ymax))) continue;
            if (xx[j - 1] == edge_x(oth, y)) {
              int off = (y == ymax) ? -1 : 1;
              float a = edge_x(cur, y + off), b = edge_x(oth, y + off);
              int v;
              if (y == cur->ymax) v = cur->dx > 0 ? RoundUp(a > b ? a : b) + 1 : RoundUp(a < b ? a : b) - 1;
              else v = cur->dx > 0 ? RoundUp(a < b ? a : b) - 1 : RoundUp(a > b ? a : b) + 1;
              bool want_left = (y == cur->ymax) ? (cur->dx > 0) : (cur->dx < 0);
              float corner_x = xx[j - 1];
              if ((float)v == corner_x || (((float)v < corner_x) == want_left)) xx[j - 1] = (float)v;
              break;
            }
          }
        }
      }
    }
    std::sort(xx, xx + j);
    if (variant == 2) {
      // "only draw each polygon pixel once" with the scan position starting at 0, and the horizontal edges of this row drawn from
      // that position: one that begins to the right of it is "after the current position" and is skipped
      int x_pos = 0;
      auto horizontal_lines = [&]() {
        for (int k = 0; k < n; k++) {
          if (e[k].ymin != y || e[k].ymin != e[k].ymax) continue;
          int xmin = e[k].xmin;
          if (x_pos < xmin) continue;
          const int xmax = e[k].xmax;
          if (x_pos > xmin) {
            xmin = x_pos;
            if (xmax < xmin) continue;
          }
          hline(im, xmin, y, xmax);
          x_pos = xmax + 1;
        }
      };
      for (int i = 1; i < j; i += 2) {
        const int x_end = RoundDown(xx[i]);
        if (x_end < x_pos) continue;
        horizontal_lines();
        if (x_end < x_pos) continue;
        int x_start = RoundUp(xx[i - 1]);
        if (x_pos > x_start) {
          x_start = x_pos;
          if (x_end < x_start) continue;
        }
        hline_swapping(im, x_start, y, x_end);
        x_pos = x_end + 1;
      }
      horizontal_lines();
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
      hline(im, x_start, y, x_end);
      x_pos = x_end + 1;
    }
  }
}

// ImagingDrawEllipse(fill) on the truncated bbox: LUT dumped from Pillow (tools/gen_ellipse_lut.py)
static inline bool draw_ellipse(Canvas& im, int x0, int y0, int x1, int y1) {
  int a = x1 - x0, b = y1 - y0;
  if (a < 0 || b < 0) return false;
  if (a > KELLIPSELUT_AMAX || b > KELLIPSELUT_AMAX) return false;
  for (int r = 0; r <= b; r++) {
    int s = kEllipseLut_data[a][b][r][0], t = kEllipseLut_data[a][b][r][1];
    if (s > t) continue;
    hline(im, x0 + s, y0 + r, x0 + t);
  }
  return true;
}

}  // namespace b2o
