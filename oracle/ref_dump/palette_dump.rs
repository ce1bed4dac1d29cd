// APPEND-TO: jxl/src/frame/modular/transforms/palette.rs
// do_palette_step_general of the reference on a random index plane: the plain gather (num_deltas == 0, Predictor::Zero),
// delta entries added to every non-weighted predictor, and the Weighted predictor with the default header.  The index
// plane holds regular, delta (index < num_deltas), implicit-cube (index >= num_colors + num_deltas) and negative
// (DELTA_PALETTE) entries.
#[cfg(test)]
mod ref_dump {
    use super::*;
    use crate::headers::bit_depth::BitDepth;
    use crate::ref_dump_io::{dir, write_i32};
    use num_traits::FromPrimitive;
    use rand::{Rng, SeedableRng};

    fn flat(i: &Image<i32>) -> Vec<i32> {
        (0..i.size().1).flat_map(|y| i.row(y).to_vec()).collect()
    }

    #[test]
    fn ref_dump_palette_steps() {
        if dir().is_none() {
            return;
        }
        let (w, h, nb) = (61usize, 37usize, 3usize);
        let (num_colors, num_deltas) = (40usize, 8usize);
        let mut rng = rand_xorshift::XorShiftRng::seed_from_u64(0);
        let depth = BitDepth::integer_samples(8);
        let wp_header = WeightedHeader {
            all_default: true,
            p1c: 16, p2c: 10, p3ca: 7, p3cb: 7, p3cc: 7, p3cd: 0, p3ce: 0,
            w0: 13, w1: 12, w2: 12, w3: 12,
        };
        // palette meta-channel: (num_colors + num_deltas) wide, nb rows (meta_apply.rs:193-198)
        let mut pal = ModularChannel::new_with_shift((num_colors + num_deltas, nb), None, depth).unwrap();
        for c in 0..nb {
            for (i, v) in pal.data.row_mut(c).iter_mut().enumerate() {
                *v = if i < num_deltas { rng.random_range(-6..7) } else { rng.random_range(0..256) };
            }
        }
        let mut idx = ModularChannel::new((w, h), depth).unwrap();
        for y in 0..h {
            for v in idx.data.row_mut(y).iter_mut() {
                *v = rng.random_range(-20..(num_colors + num_deltas + 200) as i32);
            }
        }
        write_i32("palette_index", &[h, w], &flat(&idx.data));
        write_i32("palette_table", &[nb, num_colors + num_deltas], &flat(&pal.data));
        write_i32("palette_meta", &[3], &[num_colors as i32, num_deltas as i32, 8]);
        let run = |nc: usize, nd: usize, predictor: Predictor| -> Vec<i32> {
            let mut outs: Vec<ModularChannel> = (0..nb).map(|_| ModularChannel::new((w, h), depth).unwrap()).collect();
            {
                let mut refs: Vec<&mut ModularChannel> = outs.iter_mut().collect();
                do_palette_step_general(&idx, &pal, &mut refs, nc, nd, predictor, &wp_header);
            }
            outs.iter().flat_map(|o| flat(&o.data)).collect()
        };
        // plain gather: the whole table counts as colours
        write_i32("palette_plain", &[nb, h, w], &run(num_colors + num_deltas, 0, Predictor::Zero));
        for p in 0..14u32 {
            let predictor = Predictor::from_u32(p).unwrap();
            write_i32(&format!("palette_delta_pred{p}"), &[nb, h, w], &run(num_colors, num_deltas, predictor));
        }
    }
}
