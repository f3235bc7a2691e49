//! Runs the REFERENCE's estimator, detector and densifier over the kit's .mvec clips and prints one JSON line per frame
//! (same shape as expected_clipK.json's frames).  Written against h33p/ofps @ v1; never compiled in the build image (no
//! rustc there) -- if an accessor name has drifted, the three calls below are the whole surface.
use almeida_estimator::AlmeidaEstimator;
use block_motion_detector::BlockMotionDetection;
use ofps::prelude::v1::*;

fn bits(v: impl Iterator<Item = f32>) -> String {
    v.map(|x| x.to_bits().to_string()).collect::<Vec<_>>().join(",")
}

fn main() -> Result<()> {
    let dir = std::env::args().nth(1).unwrap_or_else(|| ".".into());
    let cam = StandardCamera::new(16.0 / 9.0, 39.6 * 9.0 / 16.0);               // ofps-suite/src/app/tracking/worker.rs:445
    for clip in 0..3 {
        let mut dec = motion_loader::create_decoder(&format!("{dir}/clip{clip}.mvec"), None)?;
        let mut est = AlmeidaEstimator::default();
        for (name, prop) in est.props_mut() {
            if let ("Use ransac", PropertyMut::Bool(b)) = (name, prop) { *b = false; }   // the LSQ path is deterministic
        }
        let det = BlockMotionDetection::default();
        let mut mv = vec![];
        let mut frame = 0;
        while dec.process_frame(&mut mv, None, 0).is_ok() {
            let (q, _) = est.estimate(&mv, &cam, None)?;
            let d = det.detect_motion(&mv);
            let mut cells = vec![];
            let mut dens = |w: usize, h: usize, keep: bool| {
                let mut mf = MotionFieldDensifier::new(w, h);
                for (p, m) in mv.iter().copied() { let c = mf.add_vector(p, m); if keep { cells.push(c); } }
                bits(MotionField::from(mf).as_slice().iter().copied())
            };
            let f14 = dens(14, 14, true);
            let f60 = dens(60, 34, false);
            println!("{{\"clip\":{clip},\"frame\":{frame},\"quat\":[{},{},{},{}],\"detect_area\":{},\"detect_field\":[{}],\"cells\":[{}],\"field_14x14\":[{}],\"field_60x34\":[{}]}}",
                q.w, q.i, q.j, q.k,
                d.as_ref().map(|(a, _)| a.to_string()).unwrap_or_else(|| "null".into()),
                d.as_ref().map(|(_, f)| bits(f.as_slice().iter().copied())).unwrap_or_default(),
                cells.iter().map(|(x, y)| format!("[{x},{y}]")).collect::<Vec<_>>().join(","), f14, f60);
            mv.clear();
            frame += 1;
        }
    }
    Ok(())
}
