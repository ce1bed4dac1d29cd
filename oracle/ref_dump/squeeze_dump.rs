// APPEND-TO: jxl/src/frame/modular/transforms/squeeze.rs
// One horizontal and one vertical inverse squeeze step of the reference (odd output sizes: the tail path too),
// whole-image form (no neighbouring chunks: in_next_avg = out_prev = None).
#[cfg(test)]
mod ref_dump {
    use super::*;
    use crate::image::Rect;
    use crate::ref_dump_io::{dir, write_i32};
    use rand::{Rng, SeedableRng};

    fn random(w: usize, h: usize, lo: i32, hi: i32, rng: &mut rand_xorshift::XorShiftRng) -> Image<i32> {
        let mut img = Image::<i32>::new((w, h)).unwrap();
        for y in 0..h {
            for v in img.row_mut(y).iter_mut() {
                *v = rng.random_range(lo..hi);
            }
        }
        img
    }
    fn flat(i: &Image<i32>) -> Vec<i32> {
        (0..i.size().1).flat_map(|y| i.row(y).to_vec()).collect()
    }

    #[test]
    fn ref_dump_unsqueeze_steps() {
        if dir().is_none() {
            return;
        }
        let mut rng = rand_xorshift::XorShiftRng::seed_from_u64(0);
        // horizontal: out 41 x 19 -> avg 21 x 19, res 20 x 19
        let (ow, oh) = (41usize, 19usize);
        let avg = random(ow.div_ceil(2), oh, 0, 256, &mut rng);
        let res = random(ow / 2, oh, -9, 10, &mut rng);
        let mut out = Image::<i32>::new((ow, oh)).unwrap();
        let ra = avg.get_rect(Rect { origin: (0, 0), size: avg.size() });
        let rr = res.get_rect(Rect { origin: (0, 0), size: res.size() });
        hsqueeze_scalar(0, &ra, &rr, None, None, &mut out);
        write_i32("unsqueeze_h_avg", &[oh, ow.div_ceil(2)], &flat(&avg));
        write_i32("unsqueeze_h_res", &[oh, ow / 2], &flat(&res));
        write_i32("unsqueeze_h_out", &[oh, ow], &flat(&out));
        // vertical: out 23 x 37 -> avg 23 x 19, res 23 x 18
        let (ow, oh) = (23usize, 37usize);
        let avg = random(ow, oh.div_ceil(2), 0, 256, &mut rng);
        let res = random(ow, oh / 2, -9, 10, &mut rng);
        let mut out = Image::<i32>::new((ow, oh)).unwrap();
        let ra = avg.get_rect(Rect { origin: (0, 0), size: avg.size() });
        let rr = res.get_rect(Rect { origin: (0, 0), size: res.size() });
        vsqueeze_scalar(0, &ra, &rr, None, None, &mut out);
        write_i32("unsqueeze_v_avg", &[oh.div_ceil(2), ow], &flat(&avg));
        write_i32("unsqueeze_v_res", &[oh / 2, ow], &flat(&res));
        write_i32("unsqueeze_v_out", &[oh, ow], &flat(&out));
    }
}
