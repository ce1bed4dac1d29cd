// Shared by the ref_dump test modules (pasted at the top of each by run.sh): the .vec writer.
#[cfg(test)]
#[allow(dead_code)]
mod ref_dump_io {
    use std::io::Write;

    pub fn dir() -> Option<std::path::PathBuf> {
        let d = std::path::PathBuf::from(std::env::var("JXL_REF_DUMP_DIR").ok()?);
        std::fs::create_dir_all(&d).ok()?;
        Some(d)
    }

    pub fn write_f32(name: &str, dims: &[usize], data: &[f32]) {
        let Some(d) = dir() else { return };
        let mut f = std::fs::File::create(d.join(format!("{name}.vec"))).unwrap();
        let dims_s: Vec<String> = dims.iter().map(|x| x.to_string()).collect();
        writeln!(f, "JXLVEC1 f32 {} {}", dims.len(), dims_s.join(" ")).unwrap();
        for v in data {
            f.write_all(&v.to_le_bytes()).unwrap();
        }
    }

    pub fn write_i32(name: &str, dims: &[usize], data: &[i32]) {
        let Some(d) = dir() else { return };
        let mut f = std::fs::File::create(d.join(format!("{name}.vec"))).unwrap();
        let dims_s: Vec<String> = dims.iter().map(|x| x.to_string()).collect();
        writeln!(f, "JXLVEC1 i32 {} {}", dims.len(), dims_s.join(" ")).unwrap();
        for v in data {
            f.write_all(&v.to_le_bytes()).unwrap();
        }
    }
}
