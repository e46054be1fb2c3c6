//! Scores of fast-ssim2 0.8.0, dssim-core 3.4.0 and butteraugli 0.9.0 on the committed golden inputs
//! (`tests/golden/raw/manifest.tsv` + `<name>.ref.rgb` / `<name>.test.rgb`, packed RGB8), printed as JSON in the
//! schema of `tests/golden/scores.json`.  Each metric goes through the same calls, in the same order, as the
//! reference's wrapper for it:
//!
//!   ssimulacra2  src/metrics/ssimulacra2.rs:85-96   Vec<[u8; 3]> -> ImgVec -> compute_ssimulacra2(ref, test)
//!   dssim        src/metrics/dssim.rs:102-114,52-70 rgb8_to_dssim_image (sRGB -> linear RGBA f32, a = 1) ->
//!                                                   Dssim::new().create_image x2 -> compare -> f64::from
//!   butteraugli  src/metrics/butteraugli.rs:70-80   Vec<RGB8> -> Img -> butteraugli(ref, test, &default) -> .score
//!   psnr         src/metrics/mod.rs:312-331         (in-tree arithmetic, restated: a cross-check of the reader)
//!
//! No codec-eval dependency on purpose: the point is the three crates' own arithmetic.
use std::env;
use std::fs;
use std::path::Path;

use butteraugli::{butteraugli as butteraugli_compare, ButteraugliParams, Img, RGB8};
use dssim_core::Dssim;
use fast_ssim2::compute_ssimulacra2;
use imgref::ImgVec;
use rgb::RGBA;

fn srgb_to_linear(srgb: u8) -> f32 {
    let s = f32::from(srgb) / 255.0;
    if s <= 0.04045 {
        s / 12.92
    } else {
        ((s + 0.055) / 1.055).powf(2.4)
    }
}

fn ssimulacra2(reference: &[u8], test: &[u8], w: usize, h: usize) -> f64 {
    let r: Vec<[u8; 3]> = reference.chunks_exact(3).map(|c| [c[0], c[1], c[2]]).collect();
    let t: Vec<[u8; 3]> = test.chunks_exact(3).map(|c| [c[0], c[1], c[2]]).collect();
    let (r, t) = (ImgVec::new(r, w, h), ImgVec::new(t, w, h));
    compute_ssimulacra2(r.as_ref(), t.as_ref()).expect("compute_ssimulacra2")
}

fn dssim(reference: &[u8], test: &[u8], w: usize, h: usize) -> f64 {
    let conv = |d: &[u8]| -> ImgVec<RGBA<f32>> {
        let px: Vec<RGBA<f32>> = d
            .chunks_exact(3)
            .map(|c| RGBA { r: srgb_to_linear(c[0]), g: srgb_to_linear(c[1]), b: srgb_to_linear(c[2]), a: 1.0 })
            .collect();
        ImgVec::new(px, w, h)
    };
    let d = Dssim::new();
    let a = d.create_image(&conv(reference)).expect("create reference image");
    let b = d.create_image(&conv(test)).expect("create test image");
    let (val, _maps) = d.compare(&a, b);
    f64::from(val)
}

fn butteraugli_score(reference: &[u8], test: &[u8], w: usize, h: usize) -> f64 {
    let px = |d: &[u8]| -> Vec<RGB8> { d.chunks_exact(3).map(|c| RGB8::new(c[0], c[1], c[2])).collect() };
    let (a, b) = (Img::new(px(reference), w, h), Img::new(px(test), w, h));
    butteraugli_compare(a.as_ref(), b.as_ref(), &ButteraugliParams::default()).expect("butteraugli").score
}

fn psnr(reference: &[u8], test: &[u8], w: usize, h: usize) -> f64 {
    let mut mse_sum = 0.0f64;
    for (r, t) in reference.iter().zip(test.iter()) {
        let d = f64::from(*r) - f64::from(*t);
        mse_sum += d * d;
    }
    let mse = mse_sum / (w * h * 3) as f64;
    if mse == 0.0 { f64::INFINITY } else { 10.0 * (255.0 * 255.0 / mse).log10() }
}

fn json_f64(v: f64) -> String {
    if v.is_infinite() { "Infinity".to_string() } else { format!("{v:?}") } // {:?} is the shortest round-trip form
}

fn main() {
    let dir = env::args().nth(1).unwrap_or_else(|| "tests/golden/raw".to_string());
    let dir = Path::new(&dir);
    let manifest = fs::read_to_string(dir.join("manifest.tsv")).expect("manifest.tsv");
    let mut rows = Vec::new();
    for line in manifest.lines().filter(|l| !l.starts_with('#') && !l.trim().is_empty()) {
        let f: Vec<&str> = line.split('\t').collect();
        let (name, w, h) = (f[0], f[1].trim().parse::<usize>().unwrap(), f[2].trim().parse::<usize>().unwrap());
        let reference = fs::read(dir.join(format!("{name}.ref.rgb"))).expect("ref");
        let test = fs::read(dir.join(format!("{name}.test.rgb"))).expect("test");
        assert_eq!(reference.len(), w * h * 3, "{name}: reference size");
        assert_eq!(test.len(), w * h * 3, "{name}: test size");
        rows.push(format!(
            " \"{name}\": {{\n  \"width\": {w}, \"height\": {h},\n  \"psnr\": {},\n  \"ssimulacra2\": {},\n  \"dssim\": {},\n  \"butteraugli\": {}\n }}",
            json_f64(psnr(&reference, &test, w, h)),
            json_f64(ssimulacra2(&reference, &test, w, h)),
            json_f64(dssim(&reference, &test, w, h)),
            json_f64(butteraugli_score(&reference, &test, w, h)),
        ));
    }
    println!("{{\n \"_crates\": {{\"fast-ssim2\": \"0.8.0\", \"dssim-core\": \"3.4.0\", \"butteraugli\": \"0.9.0\"}},\n{}\n}}", rows.join(",\n"));
}
