#!/usr/bin/env python3
"""bench.py -- throughput of the BP+OSD decode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      (N > 1, one rank per GPU)

A "step" is one pass of the hot path (BP kernel, then OSD kernel on the non-converged syndromes)
over one batch of synthetic syndromes that is already resident in HBM, followed -- for N > 1 -- by
the one exchange step the path has: the gather of the corrections to rank 0 over RCCL.  Syndromes
are independent, so the batch is sharded across ranks with no other collective (weak scaling:
per-GPU batch fixed at 131072 = 2^20 / 8, i.e. BASELINE.json configs[3] at N = 8).

Workload (BASELINE.json metric / north_star): [[1922,50]] hypergraph-product code, min-sum BP with
the variable scaling factor, max_iter = n = 1922, osd_cs order 7, iid bit-flip noise p = 0.05.
Other BASELINE configs are selectable with --config (parity-test cases, not the bench line).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (bp_method, ms_scaling_factor, max_iter, osd_method, osd_order, per-GPU batch)
    "h1922_ms_cs7": ("ms", 0.0, 0, "osd_cs", 7, 131072),    # configs[3]: the metric's configuration
    "h1922_ms_osd0": ("ms", 0.0, 0, "osd0", 0, 65536),      # configs[1]
    "h1922_ps_cs60": ("ps", 0.0, 0, "osd_cs", 60, 65536),   # configs[2]
    # configs[4]: 14520 x 29524, HBM-resident kernels.  ms_scaling_factor = 0.625 is the reference harness's default
    # (css_decode_sim.py:71); with the variable factor (0) min-sum converges on < 0.1 % of these syndromes in 100 iterations
    "l29k_ms_e15": ("ms", 0.625, 100, "osd_e", 15, 1024),
}
CPU_SAMPLE = {"l29k_ms_e15": 1}  # the oracle needs ~4.5 s per elimination and ~7 ms per OSD-E candidate at this size
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)


def make_batch(H, q, B, seed, chunk=16384):
    """iid bit-flip errors e = rng.random((B, n)) < q (numpy PCG64), syndromes s = H e mod 2."""
    rng = np.random.default_rng(seed)
    m, n = H.shape
    Hc = H.tocsr().astype(np.int32)
    syn = np.empty((B, m), dtype=np.uint8)
    err = np.empty((B, n), dtype=np.uint8)
    for lo in range(0, B, chunk):
        hi = min(B, lo + chunk)
        e = rng.random((hi - lo, n)) < q
        err[lo:hi] = e
        syn[lo:hi] = (np.asarray(Hc @ e.T.astype(np.int32)) % 2).T
    return err, syn


def cpu_baseline_worker(args):
    """Decode a slice with the CPU oracle (separate process, never touches the GPU)."""
    hz_indptr, hz_indices, shape, kw, syn = args
    import scipy.sparse as sp
    from oracle import OracleDecoder

    H = sp.csr_matrix((np.ones(len(hz_indices), dtype=np.uint8), hz_indices, hz_indptr), shape=shape)
    dec = OracleDecoder(H, **kw)
    t0 = time.perf_counter()
    r = dec.decode_batch(syn, want_llr=False)
    dt = time.perf_counter() - t0
    return dt, r["osdw"], r["converged"], r["iters"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="h1922_ms_cs7", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--p", type=float, default=0.05, help="bit-flip probability q")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="syndromes timed on the CPU oracle (0 = skip; default 16384, 1 for the large code)")
    ap.add_argument("--variant", type=int, default=0, help="BP workgroup shape (0 auto, 1, 2, 4)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the final RCCL gather")
    ap.add_argument("--no-pipeline", action="store_true", help="one decoder handle, one step at a time (no overlap of consecutive steps)")
    ap.add_argument("--max-iter", type=int, default=-1, help="override max_iter (diagnostics; -1 = the config's)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    bp_method, ms, max_iter, osd_method, osd_order, B = CONFIGS[args.config]
    if args.batch:
        B = args.batch
    if args.max_iter >= 0:
        max_iter = args.max_iter
    q = args.p

    from bp_osd_amd.codes import h1922, l29k

    large = args.config.startswith("l29k")
    if args.cpu_sample < 0:
        args.cpu_sample = CPU_SAMPLE.get(args.config, 16384)
    code = l29k() if large else h1922(compute_logicals=(rank == 0))
    H = code.hz
    m, n = H.shape
    E = H.nnz
    kw = dict(error_rate=q, max_iter=max_iter, bp_method=bp_method, ms_scaling_factor=ms,
              osd_method=osd_method, osd_order=osd_order)

    nbatch = max(1, min(args.steps, 2))
    batches = [make_batch(H, q, B, seed=1000 * rank + k) for k in range(nbatch)]

    # ---- CPU baseline leg (rank 0, N = 1 only), before the GPU is initialised in this process.
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        ns = min(args.cpu_sample, B)
        sample = np.ascontiguousarray(batches[0][1][:ns])
        import threading

        stop = threading.Event()

        def heartbeat():  # the oracle is silent for minutes on the large code; keep stderr alive
            t_start = time.time()
            while not stop.wait(60.0):
                print(f"[bench] cpu baseline still running ({time.time() - t_start:.0f} s)", file=sys.stderr, flush=True)

        hb = threading.Thread(target=heartbeat, daemon=True)
        hb.start()
        dt, c_osdw, c_conv, c_it = cpu_baseline_worker((H.indptr, H.indices, H.shape, kw, sample))
        stop.set()
        cpu = dict(n=ns, dt=dt, osdw=c_osdw, conv=c_conv, iters=c_it)

    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from bp_osd_amd import BpOsdDecoder

    # Two decoder handles (each owns a HIP stream) used alternately: step k + 1 is enqueued while step k is still
    # draining -- its last max_iter = n stragglers and its OSD kernel -- so freed CUs are picked up by the next batch's
    # workgroups.  Every step is complete before the closing barrier; --no-pipeline gives one handle, one step at a time.
    ndec = 1 if args.no_pipeline else 2
    decs = [BpOsdDecoder(H, device=local_rank, **kw) for _ in range(ndec)]
    for d in decs:
        if args.variant:
            d.set_bp_variant(args.variant)
    dec = decs[0]

    dev = torch.device("cuda", local_rank)
    d_syn = [torch.from_numpy(b[1]).to(dev) for b in batches]
    d_osdw_l = [torch.empty((B, n), dtype=torch.uint8, device=dev) for _ in range(ndec)]
    d_conv_l = [torch.empty(B, dtype=torch.uint8, device=dev) for _ in range(ndec)]
    d_iters_l = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(ndec)]
    d_osdw, d_conv, d_iters = d_osdw_l[0], d_conv_l[0], d_iters_l[0]
    # the one exchange step: corrections are bit-packed on the device (8x fewer xGMI bytes), then gathered to rank 0.
    # The gather runs on torch's stream while later steps decode, so the packed rows are double-buffered and a buffer
    # is reused only after its gather has completed.
    wpr = (n + 63) // 64
    do_gather = world > 1 and not args.no_gather
    d_packed = [torch.empty((B, wpr), dtype=torch.int64, device=dev) for _ in range(2)]
    gather_done = [None, None]
    gather_list = None
    if do_gather and rank == 0:
        gather_list = [[torch.empty((B, wpr), dtype=torch.int64, device=dev) for _ in range(world)] for _ in range(2)]
    stats = {"bp_ms": [], "osd_ms": [], "iters": 0, "osd": 0}
    pending = []  # steps enqueued but not yet finalised (at most ndec)

    def finalise(k, timed):
        """Wait for step k on its handle, record its kernel times, start its gather."""
        hnd = decs[k % ndec]
        hnd.synchronize()
        if timed:
            t = hnd.last_timing()  # HIP events on the library's stream
            stats["bp_ms"].append(t["bp_ms"])
            stats["osd_ms"].append(t["osd_ms"])
            stats["iters"] += t["bp_iterations"]
            stats["osd"] += t["osd_invocations"]
        if do_gather:
            buf = k & 1
            dist.gather(d_packed[buf], gather_list[buf] if rank == 0 else None, dst=0)
            ev = torch.cuda.Event()
            ev.record()
            gather_done[buf] = ev

    def step(k, timed):
        i = k % ndec
        while len(pending) >= ndec:  # the handle (and its output buffers) of step k - ndec must be free
            finalise(*pending.pop(0))
        decs[i].decode_batch_device(d_syn[k % nbatch].data_ptr(), B, d_osdw_l[i].data_ptr(), None, None,
                                    d_conv_l[i].data_ptr(), d_iters_l[i].data_ptr(), None)
        if do_gather:
            buf = k & 1
            if gather_done[buf] is not None:
                gather_done[buf].synchronize()
            decs[i].pack_rows_device(d_osdw_l[i].data_ptr(), B, n, d_packed[buf].data_ptr())
        pending.append((k, timed))

    def drain():
        while pending:
            finalise(*pending.pop(0))

    def fence():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k, False)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k, True)
    fence()
    elapsed = time.perf_counter() - t0
    bp_ms, osd_ms, iters_tot, osd_tot = stats["bp_ms"], stats["osd_ms"], stats["iters"], stats["osd"]

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- one more decode of batch 0 for verification / LER (outside the timed region), with the OSD-0 and BP-only
    # outputs as well (css_decode_sim.py:294-295,338-339 report those error rates next to OSD-W's)
    d_osd0 = torch.empty((B, n), dtype=torch.uint8, device=dev)
    d_bp = torch.empty((B, n), dtype=torch.uint8, device=dev)
    dec.decode_batch_device(d_syn[0].data_ptr(), B, d_osdw.data_ptr(), d_osd0.data_ptr(), d_bp.data_ptr(), d_conv.data_ptr(),
                            d_iters.data_ptr(), None)
    dec.synchronize()
    t_last = dec.last_timing()

    if rank == 0:
        steps = max(args.steps, 1)
        value = world * B * steps / elapsed
        # algorithmic bytes of the dominant kernel (BP): SURVEY.md §8(d)
        bytes_per_iter = (4 * E + 2 * n) * 8
        avg_bp_ms = float(np.mean(bp_ms)) if bp_ms else float("nan")
        avg_iters = iters_tot / steps
        algo_bytes = avg_iters * bytes_per_iter + B * (m + n)
        steps_iters_scale = (algo_bytes / (t_last["bp_iterations"] * bytes_per_iter + B * (m + n))) if t_last["bp_iterations"] else 1.0
        achieved = algo_bytes / (avg_bp_ms * 1e-3) / 1e9 if avg_bp_ms > 0 else 0.0
        traffic = osd_traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(args.config, {})
                traffic = rec.get("hbm_bytes_per_launch")
                osd_traffic = rec.get("osd_hbm_bytes_per_launch")
            except Exception:
                traffic = osd_traffic = None

        # LER of this rank's shard (osdw), definitions of css_decode_sim.py:257-280 for one sector
        ler = ler0 = ler_bp = None
        if code.lz is not None:
            err0 = torch.from_numpy(batches[0][0]).to(dev)
            lz = torch.from_numpy(code.lz.astype(np.float32)).to(dev)
            fails = fails0 = fails_bp = 0
            for lo in range(0, B, 16384):
                sl = slice(lo, lo + 16384)
                logical = lambda x: (((x[sl] ^ err0[sl]).to(torch.float32) @ lz.T) % 2).sum(dim=1) > 0
                fails += int(logical(d_osdw).sum().item())
                fails0 += int(logical(d_osd0).sum().item())
                # BP-only succeeds when it converged and left no logical error (css_decode_sim.py:331-349, one sector)
                fails_bp += int((logical(d_bp) | (d_conv[sl] == 0)).sum().item())
            ler, ler0, ler_bp = fails / B, fails0 / B, fails_bp / B
        # every correction must reproduce its syndrome (checked on the device for the whole batch)
        Hd = torch.sparse_csr_tensor(torch.from_numpy(H.indptr.astype(np.int64)), torch.from_numpy(H.indices.astype(np.int64)),
                                     torch.ones(H.nnz, dtype=torch.float32), size=H.shape).to(dev)
        synd_ok = True
        for lo in range(0, B, 8192):
            got = torch.sparse.mm(Hd, d_osdw[lo:lo + 8192].to(torch.float32).T) % 2
            synd_ok = synd_ok and bool((got.T.to(torch.uint8) == d_syn[0][lo:lo + 8192]).all().item())
        conv_frac = float(d_conv.to(torch.float32).mean().item())
        it_cpu = d_iters.cpu().numpy()

        out = {
            "metric": "syndromes decoded/sec (whole node) + logical error rate, HGP [[1922,50]] p=0.05" if not large else
                      "syndromes decoded/sec (whole node), large HGP 14520x29524 (BASELINE configs[4])",
            "value": value,
            "unit": "syndromes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: " + ("[[29524,484]] HGP of a seeded (5,6)-regular 110x132 matrix, hz 14520x29524, "
                                                  if large else "[[1922,50]] HGP (31x31 circulant 1+x^2+x^5) hz 961x1922, ") +
                            f"{'min-sum' if bp_method == 'ms' else 'product-sum'} BP"
                            f"{' variable scaling' if bp_method == 'ms' and ms == 0 else (f' scaling {ms}' if bp_method == 'ms' else '')}, max_iter={max_iter or n}, "
                            f"{osd_method} order {osd_order}, iid bit-flip q={q}",
                "per_gpu_batch": B,
                "global_batch": B * world,
                "sharding": f"independent syndromes, contiguous shards x{world}" +
                            ("" if world == 1 or args.no_gather else ", RCCL gather of bit-packed corrections to rank 0"),
                "bp_variant": args.variant,
                "pipelined_steps": ndec,
            },
            "logical_error_rate": ler,
            "logical_error_rate_eb": None if ler is None else float(np.sqrt(ler * (1 - ler) / B)),
            "osd0_logical_error_rate": ler0,
            "osd0_logical_error_rate_eb": None if ler0 is None else float(np.sqrt(ler0 * (1 - ler0) / B)),
            "bp_logical_error_rate": ler_bp,
            "bp_logical_error_rate_eb": None if ler_bp is None else float(np.sqrt(ler_bp * (1 - ler_bp) / B)),
            "corrections_reproduce_syndromes": synd_ok,
            "bp_converged_fraction": conv_frac,
            "bp_iterations_mean": float(it_cpu.mean()),
            "bp_iterations_p50_p99_max": [float(np.percentile(it_cpu, 50)), float(np.percentile(it_cpu, 99)),
                                          int(it_cpu.max())],
            "osd_invocations_per_step": osd_tot / steps,
            "kernel_ms": {"bp": avg_bp_ms, "osd": float(np.mean(osd_ms)) if osd_ms else 0.0},
            # the same two kernels with nothing else on the GPU (the verification decode after the timed region); inside
            # the timed region consecutive steps overlap, which stretches the per-launch durations above
            "kernel_ms_isolated": {"bp": t_last["bp_ms"], "osd": t_last["osd_ms"]},
            "kernel_only_syndromes_per_s_per_gpu": B / ((t_last["bp_ms"] + t_last["osd_ms"]) * 1e-3),
            "roofline": {
                "kernel": "bp_large_kernel (BP message passing, messages in HBM)" if large else
                          "bp_local_kernel / bp_kernel (BP message passing, LDS- and register-resident messages)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": algo_bytes,
                "bytes_per_iteration_per_syndrome": bytes_per_iter,
                "avg_launch_ms": avg_bp_ms,
                "isolated_launch_ms": t_last["bp_ms"],
                "frac_isolated": (algo_bytes / steps_iters_scale / (t_last["bp_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_last["bp_ms"] > 0 else None,
                "note": (("algorithmic fp64 message bytes (4E+2n)*8 per executed iteration; messages stream through HBM "
                          "(1.4 MB per syndrome, far beyond LDS)") if large else
                         ("algorithmic fp64 message bytes (4E+2n)*8 per executed iteration; messages stay in "
                          "LDS / registers, so the fraction can exceed 1 and measured HBM traffic is far lower")) +
                        ("; avg_launch_ms is measured inside the timed region, where consecutive steps overlap on two streams "
                         "(rocprofv3 shows the same stretched durations); isolated_launch_ms / frac_isolated are the same kernel "
                         "alone on the GPU" if ndec > 1 else ""),
            },
        }
        if large:
            # the OSD kernel dominates this configuration; SURVEY.md §8(d) prices it at one read+write pass over the
            # packed matrix plus the sort plus the candidate sweep per invoked syndrome
            W = (n + 1 + 63) // 64
            ncand = (1 << osd_order) - 1 if osd_method == "osd_e" else 0
            osd_bytes = (osd_tot / steps) * (2 * W * 8 * m + n * 12 + ncand * ((m + 63) // 64) * 8)
            avg_osd_ms = float(np.mean(osd_ms)) if osd_ms else float("nan")
            a = osd_bytes / (avg_osd_ms * 1e-3) / 1e9 if avg_osd_ms > 0 else 0.0
            out["roofline_osd"] = {
                "kernel": "osd_large_kernel (sort + blocked GF(2) elimination + OSD-E sweep, matrix in HBM)",
                "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                "traffic": osd_traffic, "algorithmic_bytes_per_launch": osd_bytes, "avg_launch_ms": avg_osd_ms,
                "note": "the single-pass figure of SURVEY.md §8(d); a blocked elimination revisits the trailing matrix once "
                        "per group of pivot panels and its inner loop is bound by LDS table look-ups (DESIGN.md §4.5)",
            }
        if cpu is not None:
            got = d_osdw[:cpu["n"]].cpu().numpy()
            same = bool((got == cpu["osdw"]).all() and (it_cpu[:cpu["n"]] == cpu["iters"]).all())
            out["cpu_baseline"] = {
                "value": cpu["n"] / cpu["dt"],
                "unit": "syndromes/s",
                "cores": 1,
                "kind": "port",
                "sample": f"first {cpu['n']} syndromes of batch 0, decoded one at a time by oracle/bposd_oracle.c "
                          f"(single thread, {cpu['dt']:.1f} s); the reference's ldpc/Cython path is not installable here",
                "host_cores_available": len(os.sched_getaffinity(0)),
                "gpu_matches_cpu_bit_for_bit": same,
            }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
