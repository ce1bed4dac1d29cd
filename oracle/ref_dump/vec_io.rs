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

    // ---- helpers of the instrumented dumps (oracle/ref_dump/instrument.py): a name prefix per decoded file, and a
    // once-per-(prefix, site) latch so that only the first call of an instrumented function writes its vectors
    static PREFIX: std::sync::Mutex<String> = std::sync::Mutex::new(String::new());
    static DONE: std::sync::Mutex<Vec<String>> = std::sync::Mutex::new(Vec::new());

    pub fn set_prefix(p: &str) {
        *PREFIX.lock().unwrap() = p.to_string();
    }

    pub fn name(site: &str) -> String {
        format!("{}_{}", PREFIX.lock().unwrap(), site)
    }

    pub fn once(site: &str) -> bool {
        if dir().is_none() || PREFIX.lock().unwrap().is_empty() {
            return false;
        }
        let key = name(site);
        let mut done = DONE.lock().unwrap();
        if done.contains(&key) {
            return false;
        }
        done.push(key);
        true
    }

    pub fn flat_f32(i: &crate::image::Image<f32>) -> Vec<f32> {
        (0..i.size().1).flat_map(|y| i.row(y).to_vec()).collect()
    }

    pub fn flat_i32(i: &crate::image::Image<i32>) -> Vec<i32> {
        (0..i.size().1).flat_map(|y| i.row(y).to_vec()).collect()
    }

    pub fn flat_u8(i: &crate::image::Image<u8>) -> Vec<i32> {
        (0..i.size().1).flat_map(|y| i.row(y).iter().map(|v| *v as i32).collect::<Vec<_>>()).collect()
    }

    pub fn flat_i8(i: &crate::image::Image<i8>) -> Vec<i32> {
        (0..i.size().1).flat_map(|y| i.row(y).iter().map(|v| *v as i32).collect::<Vec<_>>()).collect()
    }
}
