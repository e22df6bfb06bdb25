#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; rocpd .db directories).

    tools/make_pmc_traffic.py <fetch_dir> <write_dir> <batch> <extract_passes> <match_passes> <out.json> [two_streams 0/1]

The PMC runs call bench.py with a small batch; every extraction kernel runs `extract_passes` times over `batch` frames
(previous-frame priming + warm-up + timed steps), every matching kernel `match_passes` times.  Values are the raw counter
values (KB) converted to bytes per frame; see profiles/README.md for the gfx950 calibration."""
import sqlite3, sys, glob, os, json

MATCH = {"k_knn2_batch", "k_search_init", "k_line_match"}
FETCH_CORRECTION = 2.0      # gfx950 tallies 128-byte read requests at 64 B (MI355X_MICROARCH.md, HBM section; calibrated in profiles/README.md on this library's own access patterns)


def lsd_core(batch, two_streams=False):
    """which launch form of the sequential LSD core a batch of this size runs (lines.hip: sslam_lines_extract_batch_dev; the guest form needs a core event -- the
    two-stream step -- and two rounds of its grid of 16 workgroups per compute unit)"""
    if batch <= 64: return "k_lsd_regions_cl"
    if batch < 1024: return "k_lsd_regions<true, 4>"
    if batch <= 4096: return "k_lsd_regions<false, 4>, one workgroup per frame"
    return "k_lsd_regions<false, 4>, guest form (persistent grid)" if two_streams and batch >= 8192 else "k_lsd_regions<false, 6>"

ONCE = {"k_grad_table", "k_lgamma_table", "k_probe_stream16", "k_probe_gather16"}

def totals(path):
    agg = {}
    for db in glob.glob(os.path.join(path, '**', '*.db'), recursive=True):
        c = sqlite3.connect(db)
        for name, cname, val in c.execute("select name, counter_name, counter_value from pmc_events"):
            short = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            short = short.split('<')[0]
            agg[short] = agg.get(short, 0.0) + val
    return agg

def main():
    fdir, wdir, B, ep, mp, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    two = len(sys.argv) > 7 and sys.argv[7] == "1"
    f, w = totals(fdir), totals(wdir)
    ker = {}
    for k in sorted(set(f) | set(w)):
        if k in ONCE or not k.startswith('k_'):
            continue
        frames = B * (mp if k in MATCH else ep)
        ker[k] = {"fetch_bytes_per_frame": f.get(k, 0.0) * 1024.0 / frames, "write_bytes_per_frame": w.get(k, 0.0) * 1024.0 / frames}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, the step at batch %d (%d extraction passes, %d matching passes); raw counter values, "
                         "multi-launch kernels summed; see profiles/README.md for the gfx950 calibration" % (B, ep, mp),
               "batch": B, "lsd_core": lsd_core(B, two), "fetch_correction": FETCH_CORRECTION,
               "fetch_correction_note": "multiply fetch_bytes_per_frame by this before comparing with a byte count (write side raw)",
               "kernels": ker}, open(out, 'w'), indent=1)
    for k, v in sorted(ker.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_frame"] + kv[1]["write_bytes_per_frame"])):
        print("%-20s fetch %10.0f B/frame  write %10.0f B/frame" % (k, v["fetch_bytes_per_frame"], v["write_bytes_per_frame"]))

if __name__ == "__main__":
    main()
