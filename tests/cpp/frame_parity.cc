// Compiled-language parity test of the drop-in boundary: a VarDCT frame goes through the C++ host side
// (include/jxl_hip.hpp over the C ABI of libjxl_hip.so) and through the CPU oracle (oracle/libjxlo_fused.so, test
// infrastructure), and the reconstructed planes must be equal bit for bit.  No Python, no torch: what a maintainer's
// compiled shim would link.  Built and run by tests/test_cpp_host.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "jxl_hip.hpp"
#include "synth_frame.hpp"

int main(int argc, char** argv) {
  const int w = argc > 1 ? atoi(argv[1]) : 300, h = argc > 2 ? atoi(argv[2]) : 270;
  const int epf_iters = argc > 3 ? atoi(argv[3]) : 2;
  synth::Frame F;
  if (!synth::make(w, h, epf_iters, &F)) return 2;
  const int xb = F.xb, yb = F.yb, cw = F.cw, ngroups = F.ngroups;
  const size_t stride = F.stride;
  auto &tables = F.tables;
  auto &qy = F.qy, &qx = F.qx, &qb = F.qb, &rq = F.rq, &coeffs = F.coeffs;
  auto &tmap = F.tmap, &epf = F.epf;
  auto &ytox = F.ytox, &ytob = F.ytob;
  std::vector<float>* pl = F.pl;

  // ---- device, through the C++ host side
  try {
    jxlh::Context ctx(0, 2);
    jxlh_frame_params p = jxlh::VarDctFrame::default_params((uint32_t)w, (uint32_t)h);
    p.epf_iters = (uint32_t)epf_iters;
    jxlh::VarDctFrame frame(ctx, p);
    frame.decode_hf_global(tables);
    frame.decode_lf_group(0, 0, (uint32_t)xb, (uint32_t)yb, qy.data(), qx.data(), qb.data(), (size_t)xb);
    frame.decode_hf_metadata(0, 0, (uint32_t)xb, (uint32_t)yb, tmap.data(), rq.data(), epf.data(), (size_t)xb, ytox.data(),
                             ytob.data(), (size_t)cw);
    for (int g = 0; g < ngroups; g++) frame.decode_vardct_group((uint32_t)g, &coeffs[(size_t)g * 3 * 65536], g % 2);
    frame.slot_wait(0);
    frame.slot_wait(1);
    frame.finalize_and_render();
    ctx.sync();
    std::vector<float> out[3];
    for (auto& o : out) o.resize((size_t)w * h);
    frame.read_planes(out[0].data(), out[1].data(), out[2].data());
    size_t bad = 0;
    for (int c = 0; c < 3; c++)
      for (int y = 0; y < h; y++)
        if (memcmp(&out[c][(size_t)y * w], &pl[c][(size_t)y * stride], sizeof(float) * w) != 0) bad++;
    // an invalid call must come back as an exception carrying the status
    bool threw = false;
    try {
      frame.decode_vardct_group((uint32_t)ngroups + 5, coeffs.data());
    } catch (const jxlh::Error& e) {
      threw = e.status == JXLH_ERR_INVALID_ARGUMENT;
    }
    printf("%dx%d epf_iters=%d groups=%d: %zu differing rows, error path %s\n", w, h, epf_iters, ngroups, bad,
           threw ? "ok" : "MISSING");
    return (bad == 0 && threw) ? 0 : 1;
  } catch (const std::exception& e) {
    fprintf(stderr, "device path failed: %s\n", e.what());
    return 3;
  }
}
