"""launch_alignments* on ONE small batch (BASELINE configs[1]: 100k x 150 bp): what the call costs host to host under different cuts of
the launch pipeline.  usage: h2h_small.py [cigar]"""
import sys, json
sys.path.insert(0, "."); sys.path.insert(0, "wfa-gpu_amd/bindings")
import bench, wfagpu
cigar = len(sys.argv) > 1
wl = dict(bench.WORKLOADS["cfg2c" if cigar else "cfg2"])
buf, meta = wfagpu.generate_pairs(wl["pairs"], wl["length"], wl["error"], seed=1000)
for name, cfg in (("default", {}), ("host_pack", {"host_pack": 1}), ("host_pack, 8 threads", {"host_pack": 1, "host_pack_threads": 8}),
                  ("host_pack, 6 batches", {"host_pack": 1, "batches_per_device": 6}), ("host_pack, 2 lanes", {"host_pack": 1, "lanes_per_device": 2}),
                  ("1 batch", {"batches_per_device": 1})):
    r = bench.host_to_host(buf, meta, wl, wl["max_error"], reps=12, launch_cfg=cfg)
    st = r["stages_ms"]
    print(f"{name:32s} warm {r['pageable']['warm_ms']:.2f} best {r['pageable']['best_ms']:.2f} ms | prep {st['prep_ms']} upload {st['upload_ms']} device {st['device_ms']} d2h {st['d2h_ms']} scatter {st['scatter_ms']} pack {st['host_pack_ms']} batches {st['batches']} lanes {st['lanes']}", flush=True)
