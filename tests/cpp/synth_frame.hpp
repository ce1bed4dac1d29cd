// Synthetic VarDCT frame + its oracle reconstruction for the compiled-language tests (tests/cpp/*.cc): every 4x4-block
// cell of a group is one DCT32X32, four DCT16X16 or sixteen DCT8X8; random LF, sparse coefficients, random EPF
// sharpness and colour-correlation maps.  The oracle (oracle/libjxlo_fused.so) is test infrastructure.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "jxl_hip.h"
extern "C" {
#include "jxlo.h"
}

namespace synth {
struct Rng {  // splitmix64
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  int range(int lo, int hi) { return lo + (int)(next() % (uint64_t)(hi - lo + 1)); }
  double unit() { return (double)(next() >> 11) / (double)(1ull << 53); }
};

struct Frame {
  int w, h, epf_iters, xb, yb, xg, yg, cw, ch, ngroups;
  std::vector<uint8_t> tmap, epf;
  std::vector<int32_t> rq, qy, qx, qb, coeffs;
  std::vector<int8_t> ytox, ytob;
  std::array<std::vector<float>, JXLH_NUM_QUANT_TABLES> tables;
  size_t stride;             // row pitch of the oracle planes
  std::vector<float> pl[3];  // the oracle's result (X, Y, B)
};

// returns false when the oracle refuses a table
inline bool make(int w, int h, int epf_iters, Frame* out) {
  Frame& F = *out;
  F.w = w; F.h = h; F.epf_iters = epf_iters;
  const int xb = (w + 7) / 8, yb = (h + 7) / 8, xg = (w + 255) / 256, yg = (h + 255) / 256;
  F.xb = xb; F.yb = yb; F.xg = xg; F.yg = yg;
  Rng rng{0x4A584Cull * 1000 + (uint64_t)w * 31 + h};
  std::vector<uint8_t>& tmap = F.tmap;
  std::vector<uint8_t>& epf = F.epf;
  std::vector<int32_t>& rq = F.rq;
  tmap.assign((size_t)xb * yb, 0);
  epf.resize((size_t)xb * yb);
  rq.resize((size_t)xb * yb);
  for (int by = 0; by < yb; by += 4)
    for (int bx = 0; bx < xb; bx += 4) {
      const bool fits32 = bx + 4 <= xb && by + 4 <= yb, fits16 = (xb - bx) % 2 == 0 && (yb - by) % 2 == 0;
      const int choice = rng.range(0, 2);
      const int q = rng.range(2, 16);
      auto fill = [&](int x0, int y0, int n, int type) {
        for (int y = 0; y < n; y++)
          for (int x = 0; x < n; x++) {
            tmap[(size_t)(y0 + y) * xb + x0 + x] = (uint8_t)(type | ((x == 0 && y == 0) ? 0x80 : 0));
            rq[(size_t)(y0 + y) * xb + x0 + x] = q;
          }
      };
      if (choice == 2 && fits32) {
        fill(bx, by, 4, 5);
      } else if (choice == 1 && fits32 && fits16) {
        for (int y = 0; y < 4; y += 2)
          for (int x = 0; x < 4; x += 2) fill(bx + x, by + y, 2, 4);
      } else {
        for (int y = by; y < by + 4 && y < yb; y++)
          for (int x = bx; x < bx + 4 && x < xb; x++) fill(x, y, 1, 0);
      }
    }
  for (auto& e : epf) e = (uint8_t)rng.range(0, 7);
  const int cw = (xb + 7) / 8, ch = (yb + 7) / 8;
  F.cw = cw; F.ch = ch;
  std::vector<int8_t>& ytox = F.ytox;
  std::vector<int8_t>& ytob = F.ytob;
  ytox.resize((size_t)cw * ch);
  ytob.resize((size_t)cw * ch);
  for (auto& v : ytox) v = (int8_t)rng.range(-16, 16);
  for (auto& v : ytob) v = (int8_t)rng.range(-16, 16);
  // ---- quantised LF (coded order Y, X, B) and coefficients (sparse, low frequencies denser)
  std::vector<int32_t>&qy = F.qy, &qx = F.qx, &qb = F.qb;
  qy.resize((size_t)xb * yb);
  qx.resize((size_t)xb * yb);
  qb.resize((size_t)xb * yb);
  for (size_t i = 0; i < qy.size(); i++) {
    qy[i] = rng.range(200, 900);
    qx[i] = rng.range(-30, 30);
    qb[i] = rng.range(-60, 60);
  }
  const int ngroups = xg * yg;
  F.ngroups = ngroups;
  std::vector<int32_t>& coeffs = F.coeffs;
  coeffs.assign((size_t)ngroups * 3 * 65536, 0);
  for (int g = 0; g < ngroups; g++) {
    const int gx = g % xg, gy = g / xg;
    size_t off = 0;
    for (int by = gy * 32; by < std::min(yb, gy * 32 + 32); by++)
      for (int bx = gx * 32; bx < std::min(xb, gx * 32 + 32); bx++) {
        const uint8_t t = tmap[(size_t)by * xb + bx];
        if (!(t & 0x80)) continue;
        const int n = (t & 127) == 5 ? 16 * 64 : (t & 127) == 4 ? 4 * 64 : 64;
        for (int c = 0; c < 3; c++)
          for (int k = 0; k < n; k++)
            if (rng.unit() < 0.12) coeffs[((size_t)g * 3 + c) * 65536 + off + k] = rng.range(-9, 9);
        off += n;
      }
  }

  // ---- oracle
  JxloFrameParams op;
  jxlo_default_frame_params(&op, w, h);
  op.epf_iters = epf_iters;
  std::array<std::vector<float>, JXLH_NUM_QUANT_TABLES>& tables = F.tables;
  const float* tptr[17];
  for (int t = 0; t < 17; t++) {
    tables[t].resize((size_t)jxlo_quant_table_size(t) * 3);
    if (jxlo_library_dequant_table(t, tables[t].data()) != 0) return fprintf(stderr, "oracle table %d failed\n", t), false;
    tptr[t] = tables[t].data();
  }
  std::vector<float> lf[3];
  for (auto& p : lf) p.resize((size_t)xb * yb);
  jxlo_dequant_lf(&op, qy.data(), qx.data(), qb.data(), 1.0f, (size_t)xb * yb, lf[0].data(), lf[1].data(), lf[2].data());
  const size_t stride = (size_t)xb * 8;
  F.stride = stride;
  std::vector<float>* pl = F.pl;
  std::vector<float> tm[3];
  float *plp[3], *tmpp[3], *lfp[3];
  for (int c = 0; c < 3; c++) {
    pl[c].assign(stride * yb * 8, 0.f);
    tm[c].assign(stride * yb * 8, 0.f);
    plp[c] = pl[c].data();
    tmpp[c] = tm[c].data();
    lfp[c] = lf[c].data();
  }
  jxlo_vardct_frame(&op, coeffs.data(), tmap.data(), rq.data(), epf.data(), ytox.data(), ytob.data(), lfp, tptr, plp, tmpp,
                    stride, 8);
  return true;
}
}  // namespace synth
