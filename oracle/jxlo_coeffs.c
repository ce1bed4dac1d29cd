/* CPU oracle (TEST INFRASTRUCTURE, see jxlo.h): sparse coefficient transport.
 *
 * The reference's entropy loop writes `current_coeffs[coeff_index] += coeff` into dense
 * per-group i32 slabs (jxl/src/frame/group.rs:557-572; slabs coeffs[3][65536], :437-440).  The
 * device ABI lets the host send just those (position, value) updates (include/jxl_hip.h,
 * jxlh_submit_group_sparse); this is the dense slab they describe: zero, then wrapping `+=`. */
#include <string.h>

#include "jxlo.h"

void jxlo_expand_sparse(const uint32_t* pairs, const uint32_t n[3], const uint32_t* wide, uint32_t n_wide,
                        int32_t* slab /* 3 * 65536 */) {
  memset(slab, 0, sizeof(int32_t) * 3 * 65536);
  const uint32_t* p = pairs;
  for (int c = 0; c < 3; c++) {
    uint32_t* dst = (uint32_t*)slab + (size_t)c * 65536;
    for (uint32_t i = 0; i < n[c]; i++, p++) {
      const uint32_t pos = *p & 0xffffu;
      const int32_t val = (int16_t)(*p >> 16);
      dst[pos] += (uint32_t)val; /* wrapping add, like Rust's release-mode i32 += in the reference */
    }
  }
  for (uint32_t i = 0; i < n_wide; i++) ((uint32_t*)slab)[wide[2 * i]] += wide[2 * i + 1];
}
