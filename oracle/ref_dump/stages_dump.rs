// APPEND-TO: jxl/src/render/stages/epf/test.rs
// Gaborish, EPF0, EPF1, EPF2 of the reference on a 3-channel 150x100 image (ragged against the 8x8 block grid and
// the 256-pixel group) with a variable sigma map: inputs and outputs as .vec files.
#[test]
fn ref_dump_restoration_stages() -> Result<()> {
    use crate::render::stages::GaborishStage;
    use crate::render::test::make_and_run_simple_pipeline;
    use crate::ref_dump_io::{dir, write_f32};
    if dir().is_none() {
        return Ok(());
    }
    let (w, h) = (150usize, 100usize);
    let mut rng = rand_xorshift::XorShiftRng::seed_from_u64(0);
    let mut planes: Vec<Image<f32>> = Vec::new();
    for _ in 0..3 {
        planes.push(Image::new_random((w, h), &mut rng)?);
    }
    // 1/sigma per 8x8 block, in the range the VarDCT path produces (negative; some below MIN_SIGMA -> pass-through)
    let (bw, bh) = (w.div_ceil(8) + 2, h.div_ceil(8));
    let mut sigma_img: Image<f32> = Image::new_random((bw, bh), &mut rng)?;
    for y in 0..bh {
        for v in sigma_img.row_mut(y).iter_mut() {
            *v = -0.3 - 12.0 * v.abs();
        }
    }
    let flat = |imgs: &[Image<f32>]| -> Vec<f32> {
        imgs.iter().flat_map(|i| (0..i.size().1).flat_map(|y| i.row(y).to_vec()).collect::<Vec<_>>()).collect()
    };
    write_f32("stages_input", &[3, h, w], &flat(&planes));
    write_f32("stages_inv_sigma", &[bh, bw], &flat(std::slice::from_ref(&sigma_img)));
    let sigma = Arc::new(RwLock::new(SigmaSource::Variable(Arc::new(sigma_img))));
    let scale = [40.0f32, 5.0, 3.5];
    let bsm = 2.0f32 / 3.0;
    let out0 = make_and_run_simple_pipeline(Epf0Stage::new(0.9, bsm, scale, sigma.clone()), &planes, (w, h), 0, 256)?;
    write_f32("stages_epf0", &[3, h, w], &flat(&out0));
    let out1 = make_and_run_simple_pipeline(Epf1Stage::new(1.0, bsm, scale, sigma.clone()), &planes, (w, h), 0, 256)?;
    write_f32("stages_epf1", &[3, h, w], &flat(&out1));
    let out2 = make_and_run_simple_pipeline(Epf2Stage::new(6.5, bsm, scale, sigma.clone()), &planes, (w, h), 0, 256)?;
    write_f32("stages_epf2", &[3, h, w], &flat(&out2));
    let gab = make_and_run_simple_pipeline(
        GaborishStage::new(0, 0.115169525, 0.061248592),
        std::slice::from_ref(&planes[0]),
        (w, h),
        0,
        256,
    )?;
    write_f32("stages_gaborish", &[1, h, w], &flat(&gab));
    Ok(())
}
