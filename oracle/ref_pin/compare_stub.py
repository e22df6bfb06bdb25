#!/usr/bin/env python3
"""`make -C oracle/ref_pin pin-stub`: the REFERENCE's own ORBextractor.cc -- compiled unmodified, where it lies, against the stub cv:: layer
of oracle/ref_pin/stub_cv (OpenCV leaves = oracle/cvleaf.h) -- against the CPU oracle, fixture by fixture, byte by byte.

    compare_stub.py <ref_dir (oracle/_ref)> <fixtures dir> <report.json>

Five builds of the same reference source are run:
  ref_orb_stub          -O3 -march=native -ffp-contract=off, monotonic allocator, correctly rounded cosf / sinf   = the oracle's decisions D1 + D4 + D5: MUST equal the oracle
  ref_orb_stub_fma      the same with GCC's default contraction                     -> D4's error bar (FMA in x*b + y*a, src/ORBextractor.cc:118-120)
  ref_orb_stub_malloc   the same with glibc malloc                                  -> D1's error bar (sort by heap pointer, src/ORBextractor.cc:684)
  ref_orb_stub_libm     the same with libm's own cosf / sinf                        -> D5's error bar (src/ORBextractor.cc:113; found by campaign_orb.py: one descriptor bit in 73 M)
  ref_orb_stub_asbuilt  -O3 -march=native, glibc malloc, libm  = the reference's own CMake flags (CMakeLists.txt:10-11) -> all three together
  ref_orb_stub_gauss340 the pinned binary with the stub's GaussianBlur switched to OpenCV 3.4.0's rounded taps  -> D6's error bar (descriptor bits only)
The report says, per fixture: equal or not for the pinned build; for the other three how many keypoints (as a set of (octave, x, y))
and how many descriptor bits differ from the oracle.  Exit status 0 either way -- tests/test_pin_cpu.py turns the report into a verdict."""
import glob, json, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
import oracle_lib

KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
ref, fx, report_path = sys.argv[1], sys.argv[2], sys.argv[3]
orc = oracle_lib.Oracle()
# fixture name -> extractor parameters (nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST); default = Examples/ICL.yaml:41-54
PARAMS = {"synth4lev15_700": (700, 1.5, 4, 20, 7), "synthth_1200": (1200, 1.2, 8, 35, 12)}
VARIANTS = ["ref_orb_stub", "ref_orb_stub_fma", "ref_orb_stub_malloc", "ref_orb_stub_libm", "ref_orb_stub_asbuilt", "ref_orb_stub_gauss340"]      # the last one: the pinned binary with SSLAM_STUB_GAUSS_340=1 (decision D6's alternative)


def run(binary, pgm, params):
    out = os.path.join(fx, os.path.basename(pgm)[:-4] + "." + binary)
    env = dict(os.environ); exe = binary
    if binary == "ref_orb_stub_gauss340": env["SSLAM_STUB_GAUSS_340"] = "1"; exe = "ref_orb_stub"
    subprocess.run([os.path.join(ref, exe), pgm, out] + [repr(p) if isinstance(p, float) else str(p) for p in params], check=True, capture_output=True, env=env)
    return np.fromfile(out + "_kp.bin", dtype=KP), np.fromfile(out + "_desc.bin", dtype=np.uint8).reshape(-1, 32), np.fromfile(out + "_tables.bin", dtype=np.float32)


def diff(kp, desc, okp, odesc):
    """keypoints as a set of (octave, x bits, y bits); descriptor bits over the keypoints both sides hold"""
    key = lambda k: {(int(o), x.tobytes(), y.tobytes()): i for i, (o, x, y) in enumerate(zip(k["octave"], k["x"], k["y"]))}
    a, b = key(kp), key(okp)
    common = [(a[k], b[k]) for k in a if k in b]
    bits = int(sum(np.unpackbits(desc[i] ^ odesc[j]).sum() for i, j in common)) if common else 0
    ang = int(sum(kp["angle"][i].tobytes() != okp["angle"][j].tobytes() for i, j in common))
    rows = int(sum(bool((desc[i] != odesc[j]).any()) for i, j in common))
    return {"keypoints_reference": len(a), "keypoints_oracle": len(b), "only_in_reference": len(a) - len(common), "only_in_oracle": len(b) - len(common),
            "same_order": bool(len(kp) == len(okp) and np.array_equal(kp.view(np.uint8), okp.view(np.uint8))),
            "angles_differing": ang, "descriptor_rows_differing": rows, "descriptor_bits_differing": bits, "descriptor_bits_compared": 256 * len(common)}


report = {"what": "reference src/ORBextractor.cc compiled unmodified against oracle/ref_pin/stub_cv (leaves: oracle/cvleaf.h) vs the CPU oracle",
          "fixtures": {}, "all_equal": True, "error_bars": {v: {"keypoints_differing": 0, "keypoints": 0, "descriptor_bits_differing": 0, "descriptor_bits_compared": 0} for v in VARIANTS[1:]}}
for pgm in sorted(glob.glob(os.path.join(fx, "*.pgm"))):
    name = os.path.basename(pgm)[:-4]
    if name.startswith("p") and name.count("_") == 5:      # campaign_orb.py --params: p<k>_<nfeatures>_<scaleFactor x 100>_<nlevels>_<iniThFAST>_<minThFAST>
        f_ = name.split("_"); params = (int(f_[1]), int(f_[2]) / 100.0, int(f_[3]), int(f_[4]), int(f_[5]))
    else:
        params = PARAMS.get(name, (int(name.rsplit("_", 1)[1]), 1.2, 8, 20, 7))
    with open(pgm, "rb") as f:
        assert f.readline().strip() == b"P5"; w, h = map(int, f.readline().split()); f.readline()
        img = np.frombuffer(f.read(), np.uint8).reshape(h, w)
    okp, odesc = orc.orb_extract(img, *params)
    okp = okp.view(KP).reshape(-1) if okp.dtype != KP else okp
    sc = orc.orb_params(params[0], params[1], params[2])
    res = {"params": list(params), "size": [w, h]}
    for v in VARIANTS:
        kp, desc, tables = run(v, pgm, params)
        d = diff(kp, desc, okp, odesc)
        if v == "ref_orb_stub":
            eq_kp = bool(kp.size == okp.size and np.array_equal(kp.view(np.uint8), okp.view(np.uint8)))
            eq_d = bool(desc.shape == odesc.shape and np.array_equal(desc, odesc))
            res["pinned"] = {"kp_equal": eq_kp, "desc_equal": eq_d, "keypoints": int(kp.size), **({} if eq_kp and eq_d else d)}
            report["all_equal"] &= eq_kp and eq_d
            res["scale_tables_equal"] = bool(np.array_equal(tables[:params[2]].view(np.uint32), np.asarray(sc[0], np.float32).view(np.uint32)))
            report["all_equal"] &= res["scale_tables_equal"]
        else:
            if v == "ref_orb_stub_gauss340":          # the alternative is selectable in the oracle and in the library: the reference compiled with it must equal the oracle's variant 1
                orc.set_gauss_variant(1); okp1, odesc1 = orc.orb_extract(img, *params); orc.set_gauss_variant(0)
                d["equals_oracle_variant_1"] = bool(kp.size == okp1.size and np.array_equal(kp.view(np.uint8), okp1.view(np.uint8)) and np.array_equal(desc, odesc1))
                report["all_equal"] &= d["equals_oracle_variant_1"]
            res[v] = d
            e = report["error_bars"][v]
            e["keypoints_differing"] += d["only_in_reference"] + d["only_in_oracle"]; e["keypoints"] += d["keypoints_reference"]
            e["descriptor_bits_differing"] += d["descriptor_bits_differing"]; e["descriptor_bits_compared"] += d["descriptor_bits_compared"]
    report["fixtures"][name] = res
json.dump(report, open(report_path, "w"), indent=1)
for n, r in report["fixtures"].items():
    print("%-20s pinned kp %s desc %s (%d kp) | gauss 3.4.0: %d bits differ | fma: %d kp / %d bits differ | malloc: %d kp differ | libm cosf/sinf: %d bits differ | as built: %d kp / %d bits differ" % (
        n, r["pinned"]["kp_equal"], r["pinned"]["desc_equal"], r["pinned"]["keypoints"], r["ref_orb_stub_gauss340"]["descriptor_bits_differing"],
        r["ref_orb_stub_fma"]["only_in_reference"] + r["ref_orb_stub_fma"]["only_in_oracle"], r["ref_orb_stub_fma"]["descriptor_bits_differing"],
        r["ref_orb_stub_malloc"]["only_in_reference"] + r["ref_orb_stub_malloc"]["only_in_oracle"], r["ref_orb_stub_libm"]["descriptor_bits_differing"],
        r["ref_orb_stub_asbuilt"]["only_in_reference"] + r["ref_orb_stub_asbuilt"]["only_in_oracle"], r["ref_orb_stub_asbuilt"]["descriptor_bits_differing"]))
print("reference ORBextractor.cc (stub cv) == oracle on every fixture:", report["all_equal"])
print("error bars:", json.dumps(report["error_bars"]))
