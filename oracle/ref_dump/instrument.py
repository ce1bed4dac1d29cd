#!/usr/bin/env python3
"""Instruments a SCRATCH COPY of the reference (never /root/reference itself) so that it dumps inputs and outputs of
functions that cannot be called from a unit test without a decoded frame around them:

  adaptive_lf_smoothing      jxl/src/frame/adaptive_lf_smoothing.rs:44-125
  dequant_lf                 jxl/src/frame/modular/mod.rs:837-929          (both branches: 4:4:4 and sub-sampled)
  SigmaSource::new           jxl/src/features/epf.rs:35-87
  decode_vardct_group        jxl/src/frame/group.rs:383-626                (group 0: coefficients, maps, LF, tables, pixels)

Each patch is (file, anchor text that must occur exactly once, where to insert, snippet).  The snippets are
`#[cfg(test)]` blocks calling the `ref_dump_io` module run.sh appends to jxl/src/lib.rs; frame_dump.rs decodes two
files of the reference's own test corpus to drive them.  `python instrument.py --check /root/reference` verifies
every anchor against a tree without touching it (tests/test_ref_stage_vectors.py does that on every CPU run);
`python instrument.py <scratch copy>` applies the patches.
"""
import os
import sys

PATCHES = [
    # ---------------------------------------------------------------------------------------- LF smoothing
    ("jxl/src/frame/adaptive_lf_smoothing.rs",
     "    *lf_image = [smoothed0, smoothed1, smoothed2];\n",
     "before",
     """    #[cfg(test)]
    {
        use crate::ref_dump_io::{flat_f32, name, once, write_f32};
        if once("lfs") {
            write_f32(&name("lfs_factors"), &[3], &lf_factors);
            let dims = [3, ysize, xsize];
            write_f32(&name("lfs_input"), &dims,
                      &[flat_f32(&lf_image[0]), flat_f32(&lf_image[1]), flat_f32(&lf_image[2])].concat());
            write_f32(&name("lfs_output"), &dims,
                      &[flat_f32(&smoothed0), flat_f32(&smoothed1), flat_f32(&smoothed2)].concat());
        }
    }
"""),
    # ---------------------------------------------------------------------------------------- dequant_lf, 4:4:4
    ("jxl/src/frame/modular/mod.rs",
     "                dec_row_b[x] = in_y * cfl_fac_b + in_b;\n            }\n        }\n",
     "after",
     """        #[cfg(test)]
        {
            use crate::ref_dump_io::{name, once, write_f32, write_i32};
            if once("dqlf") {
                let (w, h) = r.size;
                let mut q: Vec<i32> = Vec::new();
                for c in 0..3 {
                    for y in 0..h {
                        q.extend_from_slice(&input[c].row(y)[..w]);
                    }
                }
                write_i32(&name("dqlf_input_yxb"), &[3, h, w], &q);
                write_f32(&name("dqlf_params"), &[5], &[fac_x, fac_y, fac_b, cfl_fac_x, cfl_fac_b]);
                let mut o: Vec<f32> = Vec::new();
                for y in 0..h { o.extend_from_slice(&lf0.typed_row_mut::<f32>(y)[..w]); }
                for y in 0..h { o.extend_from_slice(&lf1.typed_row_mut::<f32>(y)[..w]); }
                for y in 0..h { o.extend_from_slice(&lf2.typed_row_mut::<f32>(y)[..w]); }
                write_f32(&name("dqlf_output_xyb"), &[3, h, w], &o);
            }
        }
"""),
    # ---------------------------------------------------------------------------------------- dequant_lf, sub-sampled
    ("jxl/src/frame/modular/mod.rs",
     "                    row[x] = *val as f32 * fac;\n                }\n            }\n",
     "after",
     """            #[cfg(test)]
            {
                use crate::ref_dump_io::{name, once, write_f32, write_i32};
                if once(&format!("dqlfsub{c}")) {
                    let (w, h) = rect_size;
                    let mut q: Vec<i32> = Vec::new();
                    let mut o: Vec<f32> = Vec::new();
                    for y in 0..h {
                        q.extend_from_slice(&ch.row(y)[..w]);
                        o.extend_from_slice(&lf[c].typed_row_mut::<f32>(y)[..w]);
                    }
                    write_i32(&name(&format!("dqlfsub{c}_input")), &[h, w], &q);
                    write_f32(&name(&format!("dqlfsub{c}_fac")), &[1], &[fac]);
                    write_f32(&name(&format!("dqlfsub{c}_output")), &[h, w], &o);
                }
            }
"""),
    # ---------------------------------------------------------------------------------------- sigma map
    ("jxl/src/features/epf.rs",
     "            Ok(SigmaSource::Variable(Arc::new(sigma_image)))\n",
     "before",
     """            #[cfg(test)]
            {
                use crate::ref_dump_io::{flat_f32, flat_i32, flat_u8, name, once, write_f32, write_i32};
                if once("sigma") {
                    let dims = [sigma_ysize, sigma_xsize];
                    write_i32(&name("sigma_raw_quant"), &dims, &flat_i32(&hf_meta.raw_quant_map));
                    write_i32(&name("sigma_transform_map"), &dims, &flat_u8(&hf_meta.transform_map));
                    write_i32(&name("sigma_epf_map"), &dims, &flat_u8(&hf_meta.epf_map));
                    let mut p = vec![rf.epf_quant_mul, quant_scale];
                    p.extend_from_slice(&rf.epf_sharp_lut);
                    write_f32(&name("sigma_params"), &[p.len()], &p);
                    write_f32(&name("sigma_inv_sigma"), &[sigma_ysize, sigma_xsize + 2], &flat_f32(&sigma_image));
                }
            }
"""),
    # ---------------------------------------------------------------------------------------- one whole group
    ("jxl/src/frame/group.rs",
     "    for PassInfo {\n        pass, br, reader, ..\n    } in pass_info.iter_mut()\n",
     "before",
     """    #[cfg(test)]
    {
        use crate::ref_dump_io::{flat_f32, name, once, write_f32, write_i32};
        if group == 0 && pixels.is_some() && once("group") {
            let (bw, bh) = block_group_rect.size;
            write_i32(&name("group_coeffs_xyb"), &[3, GROUP_DIM * GROUP_DIM],
                      &[coeffs[0].to_vec(), coeffs[1].to_vec(), coeffs[2].to_vec()].concat());
            let mut tm: Vec<i32> = Vec::new();
            let mut rq: Vec<i32> = Vec::new();
            for y in 0..bh {
                tm.extend(transform_map.row(y)[..bw].iter().map(|v| *v as i32));
                rq.extend_from_slice(&raw_quant_map.row(y)[..bw]);
            }
            write_i32(&name("group_transform_map"), &[bh, bw], &tm);
            write_i32(&name("group_raw_quant"), &[bh, bw], &rq);
            let (cw, chh) = cmap_rect.size;
            let mut yx: Vec<i32> = Vec::new();
            let mut yb: Vec<i32> = Vec::new();
            for y in 0..chh {
                yx.extend(ytox_map.row(y)[..cw].iter().map(|v| *v as i32));
                yb.extend(ytob_map.row(y)[..cw].iter().map(|v| *v as i32));
            }
            write_i32(&name("group_ytox"), &[chh, cw], &yx);
            write_i32(&name("group_ytob"), &[chh, cw], &yb);
            // the LF samples of the group's blocks (4:4:4 layout: one sample per block; a sub-sampled channel holds
            // its samples in the top-left corner of the same rectangle)
            let mut lfv: Vec<f32> = Vec::new();
            for c in 0..3 {
                let lr = lf_image[c].get_rect(block_group_rect);
                for y in 0..bh {
                    lfv.extend_from_slice(&lr.row(y)[..bw]);
                }
            }
            write_f32(&name("group_lf_xyb"), &[3, bh, bw], &lfv);
            write_f32(&name("group_params"), &[10],
                      &[quant_biases[0], quant_biases[1], quant_biases[2], quant_biases[3], x_dm_multiplier,
                        b_dm_multiplier, inv_global_scale, color_correlation_params.base_correlation_x,
                        color_correlation_params.base_correlation_b, color_correlation_params.color_factor as f32]);
            // the header fields the derived factors above come from (so a consumer can rebuild them and compare)
            let rf = &frame_header.restoration_filter;
            write_i32(&name("header_ints"), &[9],
                      &[quant_params.global_scale as i32, quant_params.quant_lf as i32, frame_header.x_qm_scale as i32,
                        frame_header.b_qm_scale as i32, color_correlation_params.color_factor as i32,
                        color_correlation_params.ytox_lf, color_correlation_params.ytob_lf, rf.epf_iters as i32,
                        rf.gab as i32]);
            let mut hf: Vec<f32> = lf_global.lf_quant.quant_factors.to_vec();
            hf.extend_from_slice(&[rf.gab_x_weight1, rf.gab_x_weight2, rf.gab_y_weight1, rf.gab_y_weight2,
                                   rf.gab_b_weight1, rf.gab_b_weight2]);
            hf.extend_from_slice(&rf.epf_sharp_lut);
            hf.extend_from_slice(&rf.epf_channel_scale);
            hf.extend_from_slice(&[rf.epf_quant_mul, rf.epf_pass0_sigma_scale, rf.epf_pass2_sigma_scale,
                                   rf.epf_border_sad_mul]);
            write_f32(&name("header_floats"), &[hf.len()], &hf);
            write_i32(&name("group_shifts"), &[6],
                      &[hshift[0] as i32, hshift[1] as i32, hshift[2] as i32, vshift[0] as i32, vshift[1] as i32,
                        vshift[2] as i32]);
            for t in 0..27usize {
                let ty = HfTransformType::from_usize(t).unwrap();
                let m = hf_global.dequant_matrices.matrix(ty, 0);
                write_f32(&name(&format!("group_table_type{t}")), &[m.len()], m);
            }
            if let Some(px) = pixels.as_ref() {
                for c in 0..3 {
                    let (w, h) = px[c].size();
                    write_f32(&name(&format!("group_pixels_c{c}")), &[h, w], &flat_f32(&px[c]));
                }
            }
        }
    }
"""),
]


def apply(root, check_only=False):
    problems = []
    by_file = {}
    for path, anchor, where, snippet in PATCHES:
        full = os.path.join(root, path)
        if not os.path.exists(full):
            problems.append(f"{path}: missing")
            continue
        text = by_file.get(full)
        if text is None:
            text = open(full).read()
        n = text.count(anchor)
        if n != 1:
            problems.append(f"{path}: anchor occurs {n} times: {anchor.strip().splitlines()[0]!r}")
            continue
        if not check_only:
            i = text.index(anchor)
            text = text[:i] + snippet + text[i:] if where == "before" else text[:i + len(anchor)] + snippet + text[i + len(anchor):]
        by_file[full] = text
    if problems:
        return problems
    if not check_only:
        for full, text in by_file.items():
            open(full, "w").write(text)
    return []


if __name__ == "__main__":
    check = "--check" in sys.argv
    args = [a for a in sys.argv[1:] if a != "--check"]
    if not args:
        sys.exit("usage: instrument.py [--check] <jxl-rs tree>")
    if not check and os.path.realpath(args[0]) == os.path.realpath("/root/reference"):
        sys.exit("refusing to modify the reference tree itself: point me at the scratch copy")
    bad = apply(args[0], check_only=check)
    for b in bad:
        print("ANCHOR PROBLEM:", b)
    sys.exit(1 if bad else 0)
