// APPEND-TO: jxl/src/frame/group.rs
// Drives the INSTRUMENTED reference (oracle/ref_dump/instrument.py inserted `#[cfg(test)]` dump blocks into
// adaptive_lf_smoothing, dequant_lf, SigmaSource::new and decode_vardct_group of the scratch copy) over real VarDCT
// files of the reference's own test corpus: every dump block writes its inputs and outputs the first time it runs.
#[cfg(test)]
mod ref_dump {
    use crate::tests::decode::decode;

    #[test]
    fn ref_dump_frame_internals() {
        if crate::ref_dump_io::dir().is_none() {
            return;
        }
        // a 4:4:4 VarDCT frame with adaptive LF smoothing, EPF and several transform types
        crate::ref_dump_io::set_prefix("vardct444");
        decode(include_bytes!("../../resources/test/green_queen_vardct_e3.jxl")).unwrap();
        // a chroma-subsampled (JPEG recompression) frame: the other dequant_lf branch, 8x8 transforms only
        crate::ref_dump_io::set_prefix("jpeg420");
        decode(include_bytes!("../../resources/test/multiple_lf_420.jxl")).unwrap();
    }
}
