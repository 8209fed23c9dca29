"""Stress of the launch_alignments* pipeline: random shapes of the call (pairs, lengths, batch size, CIGAR or score-only, -c),
random launch configuration (lanes, device slots, input pool -> ring of slots, fixed arena caps) -- every result against the checker,
every call under a watchdog (a deadlock shows as a timeout).   python scratch/soak_launch.py <seed> <iterations>"""
import ctypes as C, os, random, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wfa-gpu_amd", "bindings"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, wfagpu, oracle_lib
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = random.Random(seed)
lib = wfagpu.load()
bad = 0
t_all = time.time()
for it in range(iters):
    n = rng.choice([1, 7, 300, 3000, 20000, 150000])
    length = rng.choice([60, 150, 400, 1000]) if n >= 20000 else rng.choice([60, 400, 1000, 3000])
    err = rng.choice([0.01, 0.05, 0.12])
    buf, meta = wfagpu.generate_pairs(n, length, err, seed=rng.randrange(1 << 30), nthreads=8)
    cigar = rng.random() < 0.7
    pen = rng.choice([(2, 3, 1), (4, 6, 2), (1, 2, 1), (5, 3, 2)])
    batch = rng.choice([n, n, max(1, n // 3), max(1, n // 7), 37, 1000])
    cfg = dict(lanes_per_device=rng.choice([0, 1, 2, 3, 4]), virtual_devices=rng.choice([0, 0, 2, 3, 8]),
               input_pool_bytes=rng.choice([0, 0, 1 << 16, 1 << 24]), arena_limit_bytes=rng.choice([0, 0, 64 << 20]),
               batches_per_device=rng.choice([0, 0, 4, 24]), numa_pin=rng.choice([0, 1]),
               host_pack=rng.choice([0, 1, 1, -1]), host_pack_threads=rng.choice([0, 1, 3, 8]))
    if rng.random() < 0.3:
        # a few bytes outside ACGT: their batches go up as ASCII, their pairs run the byte-compare kernels
        for _ in range(rng.choice([1, 3, 40])):
            m = meta[rng.randrange(n)]
            if int(m["pattern_len"]):
                buf[int(m["pattern_offset"]) + rng.randrange(int(m["pattern_len"]))] = ord("N")
    wfagpu.configure_launch(**cfg)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=cigar, nthreads=16)
    res = C.POINTER(wfagpu.AlignmentResult)()
    assert lib.initialize_wfa_results(C.byref(res), n, rng.choice([1, 64, 512]))
    opt = wfagpu.Options(max_error=rng.choice([20, 300, int(length * 0.4)]), threads_per_block=64, num_workers=0, band=-1,
                         batch_size=batch, num_alignments=n, penalties=wfagpu.Penalties(*pen), compute_cigar=cigar)
    fn = lib.launch_alignments if cigar else lib.launch_alignments_distance
    check = rng.random() < 0.15 and n <= 20000
    m2 = meta.copy()
    done = threading.Event()
    def call():
        fn(buf.ctypes.data, buf.nbytes, m2.ctypes.data, res, opt, check)
        done.set()
    th = threading.Thread(target=call, daemon=True); th.start()
    if not done.wait(120):
        print("TIMEOUT it", it, n, length, cigar, pen, batch, cfg, flush=True); os._exit(3)
    s = np.array([res[i].error for i in range(n)], dtype=np.int64)
    ok = np.array_equal(s, so)
    if ok and cigar:
        ok = all(C.string_at(res[i].cigar.buffer).decode() == co[i] for i in range(0, n, max(1, n // 5000)))
    if check and lib.wfagpu_amd_check_failures() != 0:
        ok = False
    if not ok:
        bad += 1
        print("MISMATCH it", it, n, length, err, cigar, pen, batch, cfg, flush=True)
    lib.destroy_wfa_results(res, n)
    if rng.random() < 0.2:
        lib.wfagpu_amd_release_cache()
wfagpu.configure_launch()
print("launch soak seed", seed, "iterations", iters, "mismatching calls", bad, "%.1f s" % (time.time() - t_all))
sys.exit(1 if bad else 0)
