// APPEND-TO: jxl/src/frame/modular/transforms/rct.rs
// Every RCT op of the reference on three random i32 planes (the permutations are buffer swaps, not arithmetic).
#[cfg(test)]
mod ref_dump {
    use super::*;
    use crate::ref_dump_io::{dir, write_i32};
    use rand::{Rng, SeedableRng};

    #[test]
    fn ref_dump_rct_ops() {
        if dir().is_none() {
            return;
        }
        let (w, h) = (67usize, 33usize);
        let mut rng = rand_xorshift::XorShiftRng::seed_from_u64(0);
        let mut base = Vec::new();
        for _ in 0..3 {
            let mut img = Image::<i32>::new((w, h)).unwrap();
            for y in 0..h {
                for v in img.row_mut(y).iter_mut() {
                    *v = rng.random_range(-70000..70000);
                }
            }
            base.push(img);
        }
        let flat = |imgs: &[Image<i32>]| -> Vec<i32> {
            imgs.iter().flat_map(|i| (0..h).flat_map(|y| i.row(y).to_vec()).collect::<Vec<_>>()).collect()
        };
        write_i32("rct_input", &[3, h, w], &flat(&base));
        let ops = [
            RctOp::Noop,
            RctOp::AddFirstToThird,
            RctOp::AddFirstToSecond,
            RctOp::AddFirstToSecondAndThird,
            RctOp::AddAvgToSecond,
            RctOp::AddFirstToThirdAndAvgToSecond,
            RctOp::YCoCg,
        ];
        for (i, op) in ops.into_iter().enumerate() {
            let mut p: Vec<Image<i32>> = base.iter().map(|x| x.try_clone().unwrap()).collect();
            let (a, rest) = p.split_at_mut(1);
            let (b, c) = rest.split_at_mut(1);
            rct_loop(&mut a[0], &mut b[0], &mut c[0], op);
            write_i32(&format!("rct_op{i}"), &[3, h, w], &flat(&p));
        }
    }
}
