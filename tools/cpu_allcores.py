"""The CPU oracle on every host core at once: N independent single-threaded processes, one pinned per core, each running the bench
workload (ORB + LSD/LBD extract + matches against the previous frame) for a fixed time; prints one JSON line with the summed rate
("N independent processes pinned 1/core for fps_allcores").  Run by bench.py's cpu_baseline leg in a subprocess.
usage: python tools/cpu_allcores.py <w> <h> <nfeat> <nlines> <seconds> [max processes]"""
import json, multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(core, w, h, nfeat, nlines, seconds, q):
    try:
        os.sched_setaffinity(0, {core})
    except Exception:
        pass
    import numpy as np
    import oracle_lib
    from synth import synth_frame, warp_prev
    orc = oracle_lib.Oracle(native=True)      # built once by the parent (bench.py) before the workers start
    frames = [synth_frame(2000 + i, w, h) for i in range(4)]
    prev = warp_prev(frames[0])
    pk, pd = orc.orb_extract(prev, nfeat)
    pl = orc.lines_extract(prev, nlines) if nlines > 0 else None
    n = 0; t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        cur = frames[n % len(frames)]
        kp, d = orc.orb_extract(cur, nfeat)
        pm = np.stack([pk["x"], pk["y"]], axis=1).astype(np.float32)
        orc.search_for_initialization(pk, pd, kp, d, pm, 100, 0.9, True, (0.0, float(w), 0.0, float(h)))
        orc.knn2(pd, d)
        if nlines > 0:
            kl, ld, fn, raw = orc.lines_extract(cur, nlines)
            orc.line_match(pl[1], ld, 0.5, False)
        n += 1
    q.put((n, time.perf_counter() - t0))


def main():
    w, h, nfeat, nlines, seconds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
    try:
        cores = sorted(os.sched_getaffinity(0))
    except Exception:
        cores = list(range(os.cpu_count() or 1))
    if len(sys.argv) > 6:
        cores = cores[:int(sys.argv[6])]
    q = mp.Queue()
    procs = [mp.Process(target=worker, args=(c, w, h, nfeat, nlines, seconds, q)) for c in cores]
    for p in procs: p.start()
    res = [q.get(timeout=seconds * 4 + 60) for _ in procs]
    for p in procs: p.join(timeout=30)
    print(json.dumps({"value": sum(n / dt for n, dt in res), "unit": "frames/s", "cores": len(cores), "frames": sum(n for n, _ in res), "seconds": seconds}))


if __name__ == "__main__":
    main()
