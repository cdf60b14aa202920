#!/usr/bin/env python3
"""bench.py -- throughput of the BP+OSD decode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      (N > 1, one rank per GPU)

A "step" is one pass of the hot path (BP kernel, then OSD kernel on the non-converged syndromes) over one batch of
synthetic syndromes that is already resident in HBM, producing every output the reference's decode produces
(osdw, osd0, bp decodings, converge flag, iteration count), followed -- for N > 1 -- by the one exchange step the path
has: the gather of the bit-packed corrections to rank 0 over RCCL.  Syndromes are independent, so the batch is sharded
across ranks with no other collective (weak scaling: per-GPU batch fixed at 131072 = 2^20 / 8, i.e. BASELINE.json
configs[3] at N = 8).  The step loop is bp_osd_amd.sharding.StepPipeline (the loop tests/test_sharding_cpu.py drives on
gloo); consecutive steps overlap on the decoder handle's two lanes (HIP streams), --no-pipeline serialises them.

Workload (BASELINE.json metric / north_star): [[1922,50]] hypergraph-product code, min-sum BP with the variable scaling
factor, max_iter = n = 1922, osd_cs order 7, iid bit-flip noise p = 0.05.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the fields).  Its top-level fields are the headline
configuration.  Without --config (and N = 1) the line also carries, under "configs", a short run of every other
BASELINE.json configuration that fits one GPU -- configs[1] h1922_ms_osd0, configs[2] h1922_ps_cs60 (as the reference
computes it) and h1922_ps_cs60_clip20, configs[4] l29k_ms_e15 -- and the headline at B = 2^20 on the one GPU (configs[3]'s
whole batch): value, ms_per_step, kernel times, the BINDING roofline of the dominant kernel, the cross-kernel check.
--config NAME runs that one configuration alone with its full CPU legs (the numbers of record under profiles/).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (bp_method, ms_scaling_factor, max_iter, osd_method, osd_order, per-GPU batch, ps_clip)   [code: by name prefix]
    "h1922_ms_cs7": ("ms", 0.0, 0, "osd_cs", 7, 131072, 0.0),    # configs[3]: the metric's configuration
    # the reference's own stated workload, /root/reference/examples/qldpc_decode_example.py:5-23: [[400,16,6]] =
    # hgp(mkmn_16_4_6), min-sum with the variable scaling factor, max_iter = n, osd_cs order 42, error rate 0.05
    "hgp400_ms_cs42": ("ms", 0.0, 0, "osd_cs", 42, 131072, 0.0),
    # the same script's settings on the reference's two larger example codes (examples/codes/classical_seed_codes/mkmn_20_5_8.txt,
    # mkmn_24_6_10.txt -> [[625,25,8]], [[900,36,10]])
    "hgp625_ms_cs42": ("ms", 0.0, 0, "osd_cs", 42, 131072, 0.0),
    "hgp900_ms_cs42": ("ms", 0.0, 0, "osd_cs", 42, 131072, 0.0),
    "h1922_ms_osd0": ("ms", 0.0, 0, "osd0", 0, 65536, 0.0),      # configs[1]
    # configs[2] as the reference computes it: product-sum without clipping saturates (tanh -> 1, log -> inf, NaN)
    "h1922_ps_cs60": ("ps", 0.0, 0, "osd_cs", 60, 65536, 0.0),
    # configs[2] with the build-owned switch ps_clip = 20 (check->bit messages clamped): the numerically live variant
    "h1922_ps_cs60_clip20": ("ps", 0.0, 0, "osd_cs", 60, 65536, 20.0),
    # configs[4]: 14520 x 29524, HBM-resident kernels.  ms_scaling_factor = 0.625 is the reference harness's default
    # (css_decode_sim.py:71); with the variable factor (0) min-sum converges on < 0.1 % of these syndromes in 100 iterations
    "l29k_ms_e15": ("ms", 0.625, 100, "osd_e", 15, 1024, 0.0),
}
# what the default run adds under "configs" (name in the line -> CONFIGS key, per-GPU batch, where the batch comes from)
EXTRA_RUNS = (
    ("h1922_ms_osd0", "h1922_ms_osd0", 0, "numpy"),
    ("h1922_ps_cs60", "h1922_ps_cs60", 0, "numpy"),
    ("h1922_ps_cs60_clip20", "h1922_ps_cs60_clip20", 0, "numpy"),
    ("l29k_ms_e15", "l29k_ms_e15", 0, "numpy"),
    ("h1922_ms_cs7_b1048576", "h1922_ms_cs7", 1 << 20, "device"),
)
# syndromes timed on ONE CPU thread / per worker of the all-cores leg (the oracle needs ~1 ms per H1922 syndrome,
# ~4.5 s per L29k elimination plus ~7 ms per OSD-E candidate)
# (large code: a quarter of the syndromes go through a ~4.5-minute OSD-E sweep on one core.  One thread decodes syndromes 0..3
# of batch 0 -- one of them needs OSD --, the all-cores leg one syndrome per worker from syndrome 20 on: with 16 workers six of
# those need OSD, at most one per worker; ~12 minutes of CPU in all, which is why this is not the default bench line)
CPU_SAMPLE = {"l29k_ms_e15": (4, 1), "h1922_ps_cs60": (512, 128), "h1922_ps_cs60_clip20": (1024, 256),
              "hgp625_ms_cs42": (4096, 1024), "hgp900_ms_cs42": (2048, 512)}
# ... and in the default run's short extra legs (one core, < 10 s each; none for the large code)
CPU_SAMPLE_EXTRA = {"h1922_ms_osd0": 2048, "h1922_ps_cs60": 96, "h1922_ps_cs60_clip20": 256, "l29k_ms_e15": 0}
CPU_ALL_CORES_OFFSET = {"l29k_ms_e15": 20}  # first syndrome of the all-cores leg (default: right behind the one-thread sample)
CPU_SAMPLE_DEFAULT = (8192, 2048)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)
CLOCK_HZ = 2.4e9        # peak shader clock (MI355X_MICROARCH.md, chip-level parameters)
LDS_READ_B64_BYTES_PER_CLK = 256.0        # ds_read_b64, per CU (MI355X_MICROARCH.md, LDS table)
LDS_WRITE_B64_BYTES_PER_CLK = 512.0 / 6   # ds_write_b64: 6 cycles per wave-instruction of 512 bytes

_BATCH_CACHE = {}  # (code key, q, seed) -> (err, syn): a prefix of a larger batch of the same stream IS the smaller batch


def make_batch(H, q, B, seed, chunk=16384, cache_key=None):
    """iid bit-flip errors e = rng.random((B, n)) < q (numpy PCG64), syndromes s = H e mod 2.  The generator is consumed row
    by row, so the first B rows of a longer batch of the same seed are exactly the batch of B rows."""
    if cache_key is not None:
        hit = _BATCH_CACHE.get((cache_key, q, seed))
        if hit is not None and hit[0].shape[0] >= B:
            return hit[0][:B], hit[1][:B]
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
    if cache_key is not None:
        _BATCH_CACHE[(cache_key, q, seed)] = (err, syn)
    return err, syn


def usable_cores():
    """Host cores this process may use: the affinity mask, cut down to the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline_worker(args):
    """Decode a slice with the CPU oracle (in this process for the one-thread leg, in forked workers for the
    all-cores leg; both run BEFORE the GPU is initialised and never touch it)."""
    hz_indptr, hz_indices, shape, kw, syn, ps_math = args
    import scipy.sparse as sp
    from oracle import OracleDecoder

    H = sp.csr_matrix((np.ones(len(hz_indices), dtype=np.uint8), hz_indices, hz_indptr), shape=shape)
    okw = {k: v for k, v in kw.items() if k != "ps_math_form"}
    dec = OracleDecoder(H, ps_math=ps_math, **okw)
    t0 = time.perf_counter()
    r = dec.decode_batch(syn, want_llr=False)
    dt = time.perf_counter() - t0
    return dt, r["osdw"], r["converged"], r["iters"]


def prepare(a, rank, world):
    """Everything of one configuration that happens BEFORE the GPU is initialised in this process: the code, the decoder
    options, the seeded batches, the CPU baseline legs (rank 0, N = 1)."""
    bp_method, ms, max_iter, osd_method, osd_order, B, ps_clip = CONFIGS[a.config]
    if a.batch:
        B = a.batch
    if a.max_iter >= 0:
        max_iter = a.max_iter
    q = a.p
    from bp_osd_amd.codes import h1922, l29k, hgp

    large = a.config.startswith("l29k")
    ref400 = a.config[:6] in ("hgp400", "hgp625", "hgp900")
    ref_seed = {"hgp400": ("mkmn_16_4_6.txt", "[[400,16,6]]", "192x400"), "hgp625": ("mkmn_20_5_8.txt", "[[625,25,8]]", "300x625"),
                "hgp900": ("mkmn_24_6_10.txt", "[[900,36,10]]", "432x900")}.get(a.config[:6])
    cpu_one, cpu_per_proc = CPU_SAMPLE.get(a.config, CPU_SAMPLE_DEFAULT)
    if a.cpu_sample >= 0:
        cpu_one = a.cpu_sample
    # logical operators: the reference's generic route for H1922, the closed-form product basis for the large code
    if ref400:
        seed = np.loadtxt(os.path.join(ROOT, "tests", "golden", ref_seed[0]), dtype=int).astype(np.uint8)
        code = hgp(seed, compute_logicals=(rank == 0))
    else:
        code = l29k(compute_logicals="closed_form" if rank == 0 else False) if large else h1922(compute_logicals=(rank == 0))
    H = code.hz
    kw = dict(error_rate=q, max_iter=max_iter, bp_method=bp_method, ms_scaling_factor=ms,
              osd_method=osd_method, osd_order=osd_order, ps_clip=ps_clip)
    if bp_method == "ps":
        kw["ps_math_form"] = int(a.ps_math_form)
    P = dict(name=a.label or a.config, a=a, B=B, q=q, code=code, H=H, kw=kw, large=large, ref400=ref400, ref_seed=ref_seed,
             bp_method=bp_method, ms=ms, max_iter=max_iter, osd_method=osd_method, osd_order=osd_order, ps_clip=ps_clip,
             cpu=None, cpu_all=None, cpu_pm=None, batches=None, nbatch=max(1, min(a.steps, a.nbatch)))
    if a.gen == "device":
        return P  # the batch is drawn on the GPU (torch's generator); no CPU leg can see it
    ckey = "l29k" if large else (a.config[:6] if ref400 else "h1922")
    P["batches"] = [make_batch(H, q, B, seed=1000 * rank + k, cache_key=ckey) for k in range(P["nbatch"])]

    # ---- CPU baseline legs (rank 0, N = 1 only), before the GPU is initialised in this process.
    if rank == 0 and world == 1 and cpu_one > 0:
        ns = min(cpu_one, B)
        sample = np.ascontiguousarray(P["batches"][0][1][:ns])
        import threading

        stop = threading.Event()

        def heartbeat():  # the oracle is silent for minutes on the large code; keep stderr alive
            t_start = time.time()
            while not stop.wait(60.0):
                print(f"[bench] cpu baseline still running ({time.time() - t_start:.0f} s)", file=sys.stderr, flush=True)

        hb = threading.Thread(target=heartbeat, daemon=True)
        hb.start()
        # product-sum: the platform libm is what the reference calls -- the timed CPU leg and the "bit for bit" field use it
        # (ps_math = 0); the oracle's portable-math mode, in which the GPU is bit-exact by construction of the shared header,
        # is a separate, smaller leg reported under its own name
        dt, c_osdw, c_conv, c_it = cpu_baseline_worker((H.indptr, H.indices, H.shape, kw, sample, 0))
        P["cpu"] = dict(n=ns, dt=dt, osdw=c_osdw, conv=c_conv, iters=c_it)
        if bp_method == "ps":
            npm = min(ns, max(32, ns // 2))
            dtp, p_osdw, _, p_it = cpu_baseline_worker((H.indptr, H.indices, H.shape, kw, sample[:npm], 2 - int(a.ps_math_form)))
            P["cpu_pm"] = dict(n=npm, dt=dtp, osdw=p_osdw, iters=p_it)
        # all host cores: one process per core over disjoint shards of the same batch (the reference's execution model
        # is one decode at a time per process, css_decode_sim.py:519-520)
        procs = a.cpu_procs if a.cpu_procs >= 0 else min(usable_cores(), 64)
        lo_all = max(ns, CPU_ALL_CORES_OFFSET.get(a.config, ns))
        per = min(cpu_per_proc, max(0, (B - lo_all)) // max(procs, 1))
        if procs > 1 and per > 0:
            import multiprocessing as mp

            shards = [np.ascontiguousarray(P["batches"][0][1][lo_all + i * per: lo_all + (i + 1) * per]) for i in range(procs)]
            with mp.get_context("fork").Pool(procs) as pool:
                t0 = time.perf_counter()
                res = pool.map(cpu_baseline_worker, [(H.indptr, H.indices, H.shape, kw, s, 0) for s in shards])
                wall = time.perf_counter() - t0
            P["cpu_all"] = dict(procs=procs, per=per, wall=wall, lo=lo_all, osdw=np.concatenate([r[1] for r in res]),
                                iters=np.concatenate([r[3] for r in res]), busy=max(r[0] for r in res))
        stop.set()
    return P


def measure(P, rank, local_rank, world):
    """The GPU part of one configuration: timed steps, the verification decode, the cross-kernel check, the host-to-host leg,
    and (rank 0) the record."""
    import torch
    import torch.distributed as dist

    from bp_osd_amd import BpOsdDecoder
    from bp_osd_amd.sharding import StepPipeline

    a, B, q, code, H, kw = P["a"], P["B"], P["q"], P["code"], P["H"], P["kw"]
    large, ref400, ref_seed = P["large"], P["ref400"], P["ref_seed"]
    bp_method, ms, max_iter, osd_method, osd_order, ps_clip = (P[k] for k in ("bp_method", "ms", "max_iter", "osd_method", "osd_order", "ps_clip"))
    m, n = H.shape
    E = H.nnz
    nbatch = P["nbatch"]
    batches = P["batches"]
    cpu, cpu_all, cpu_pm = P["cpu"], P["cpu_all"], P["cpu_pm"]
    dev = torch.device("cuda", local_rank)

    # One decoder handle.  Its two lanes (HIP streams with their own workspaces) alternate between consecutive calls:
    # step k + 1 is enqueued while step k is still draining its last max_iter = n stragglers and its OSD kernel, so
    # freed CUs are picked up by the next batch's workgroups.  Every step is complete before the closing barrier.
    dec = BpOsdDecoder(H, device=local_rank, **kw)
    if a.variant:
        dec.set_bp_variant(a.variant)
    # steps in flight: two; as many as the handle has lanes on the HBM-resident path (three: see bposd_create)
    nslots = 1 if a.no_pipeline else (max(2, min(3, dec.num_lanes)) if large else 2)
    if a.slots and not a.no_pipeline:
        nslots = max(1, min(a.slots, dec.num_lanes))

    if batches is not None:
        d_syn = [torch.from_numpy(b[1]).to(dev) for b in batches]
        d_err0 = None
    else:
        # drawn on the device: e = rand(B, n) < q from torch's generator (seed 1000 * rank + k), s = H e mod 2 by a sparse product
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            Hd_gen = torch.sparse_csr_tensor(torch.from_numpy(H.indptr.astype(np.int64)), torch.from_numpy(H.indices.astype(np.int64)),
                                             torch.ones(H.nnz, dtype=torch.float32), size=H.shape).to(dev)
        d_syn, d_err0 = [], None
        for k in range(nbatch):
            gen = torch.Generator(device=dev)
            gen.manual_seed(1000 * rank + k)
            errk = torch.empty((B, n), dtype=torch.uint8, device=dev)
            synk = torch.empty((B, m), dtype=torch.uint8, device=dev)
            for lo in range(0, B, 16384):
                e = torch.rand((min(16384, B - lo), n), generator=gen, device=dev) < q
                errk[lo:lo + 16384] = e
                synk[lo:lo + 16384] = (torch.sparse.mm(Hd_gen, e.to(torch.float32).T) % 2).T.to(torch.uint8)
            d_syn.append(synk)
            if k == 0:
                d_err0 = errk
            else:
                del errk
        del Hd_gen
    mk = lambda *shape, dtype=torch.uint8: torch.empty(shape, dtype=dtype, device=dev)
    outs = [dict(osdw=mk(B, n), osd0=mk(B, n), bp=mk(B, n), conv=mk(B), iters=mk(B, dtype=torch.int32)) for _ in range(nslots)]
    # the one exchange step: corrections are bit-packed on the device (8x fewer xGMI bytes), then gathered to rank 0
    wpr = (n + 63) // 64
    do_gather = (world > 1 and not a.no_gather) or a.force_gather
    # nslots + 2 packed buffers: the gather of step k is issued when step k + nslots is about to be launched, and a launch
    # writes its packed rows itself -- with a pair, every launch waited for the gather issued a moment before (sharding.py)
    nbuf = nslots + 2
    d_packed = [torch.empty((B, wpr), dtype=torch.int64, device=dev) for _ in range(nbuf)] if do_gather else None
    # (rehearsal on one GPU: gloo moves host tensors, so the packed rows are copied to the host before the gather)
    x_packed = [torch.empty((B, wpr), dtype=torch.int64) for _ in range(nbuf)] if (do_gather and a.rehearse_on_one_gpu) else d_packed
    stats = {"bp_ms": [], "osd_ms": [], "iters": 0, "osd": 0}
    lane_of_slot = {}
    # N > 1 with the gather: where the kernels write bit-packed rows themselves (every small-path code) the decode fills the
    # gather's buffer directly -- packed syndromes in, osdw / osd0 / bp out as 64-bit words -- and no pack kernel runs between
    # the decode and the gather; on the HBM-resident path the byte rows are packed by bposd_pack_rows_device as before
    native_gather = False
    if do_gather:
        try:
            if batches is not None:
                d_psyn = [torch.from_numpy(np.concatenate([dec.pack_rows(b[1][lo:lo + 16384]) for lo in range(0, B, 16384)]).view(np.int64)).to(dev)
                          for b in batches]
            else:
                d_psyn = []
                for s in d_syn:
                    w = torch.empty((B, (m + 63) // 64), dtype=torch.int64, device=dev)
                    dec.pack_rows_device(s.data_ptr(), B, m, w.data_ptr())
                    dec.synchronize()
                    d_psyn.append(w)
            pouts = [dict(osd0=torch.empty((B, wpr), dtype=torch.int64, device=dev), bp=torch.empty((B, wpr), dtype=torch.int64, device=dev))
                     for _ in range(nslots)]
            dec.decode_batch_device_packed(d_psyn[0].data_ptr(), B, d_packed[0].data_ptr(), pouts[0]["osd0"].data_ptr(), pouts[0]["bp"].data_ptr(),
                                           outs[0]["conv"].data_ptr(), outs[0]["iters"].data_ptr())
            dec.synchronize()
            native_gather = True
        except Exception as e:  # (ValueError = BPOSD_ERR_UNSUPPORTED on the HBM-resident path; anything else: fall back as well, say so)
            if not isinstance(e, ValueError):
                print(f"[bench] native packed decode unavailable ({type(e).__name__}: {e}); packing with bposd_pack_rows_device", file=sys.stderr, flush=True)
            native_gather = False

    def launch(k, slot, buf):
        o = outs[slot]
        if native_gather:
            dec.decode_batch_device_packed(d_psyn[k % nbatch].data_ptr(), B, d_packed[buf].data_ptr(), pouts[slot]["osd0"].data_ptr(),
                                           pouts[slot]["bp"].data_ptr(), o["conv"].data_ptr(), o["iters"].data_ptr())
        else:
            dec.decode_batch_device(d_syn[k % nbatch].data_ptr(), B, o["osdw"].data_ptr(), o["osd0"].data_ptr(), o["bp"].data_ptr(),
                                    o["conv"].data_ptr(), o["iters"].data_ptr(), None)
        lane_of_slot[slot] = dec.last_lane

    def pack(slot, buf):  # queued on the lane of the decode just launched
        if not native_gather:
            dec.pack_rows_device(outs[slot]["osdw"].data_ptr(), B, n, d_packed[buf].data_ptr())

    def wait(slot):
        dec.synchronize(lane_of_slot[slot])

    def on_finalised(k, timed):
        if do_gather and a.rehearse_on_one_gpu:
            x_packed[k % nbuf].copy_(d_packed[k % nbuf])
        if timed:
            t = dec.lane_timing(lane_of_slot[k % nslots])  # HIP events on the lane's own stream
            stats["bp_ms"].append(t["bp_ms"])
            stats["osd_ms"].append(t["osd_ms"])
            stats["iters"] += t["bp_iterations"]
            stats["osd"] += t["osd_invocations"]

    pipe = StepPipeline(nslots, launch, wait, pack=pack, packed=x_packed, rows=B, gather=("force" if a.force_gather else do_gather),
                        on_finalised=on_finalised)
    for k in range(a.warmup):
        pipe.step(k, False)
    pipe.fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        pipe.step(a.warmup + k, True)
    pipe.fence()
    elapsed = time.perf_counter() - t0
    bp_ms, osd_ms, iters_tot, osd_tot = stats["bp_ms"], stats["osd_ms"], stats["iters"], stats["osd"]

    per_rank = None
    gather_ms = None
    if do_gather:
        gms = pipe.gather_ms()
        gather_ms = float(np.mean(gms)) if gms else 0.0
    if world > 1:
        # per-rank step time, BP / OSD kernel time and gather time, so that a sub-linear scaling curve can be attributed
        mine = torch.tensor([1e3 * elapsed / max(a.steps, 1), float(np.mean(bp_ms)) if bp_ms else 0.0,
                             float(np.mean(osd_ms)) if osd_ms else 0.0, gather_ms or 0.0, float(pipe.launch_waited_for_gather)],
                            dtype=torch.float64, device="cpu" if a.rehearse_on_one_gpu else dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [dict(rank=r, ms_per_step=float(t[0]), bp_ms=float(t[1]), osd_ms=float(t[2]), gather_ms=float(t[3]),
                         launches_that_waited_for_a_gather=int(t[4])) for r, t in enumerate(allr)]
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if a.rehearse_on_one_gpu else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    elif a.force_gather:
        # world size 1 on the RCCL backend: one more collective on a device tensor (the LER counters' all-reduce of the N > 1 job)
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- one more decode of batch 0 for verification / LER (outside the timed region, nothing else on the GPU: these are
    # the "isolated" kernel durations)
    o = outs[0]
    d_osdw, d_osd0, d_bp, d_conv, d_iters = o["osdw"], o["osd0"], o["bp"], o["conv"], o["iters"]
    dec.decode_batch_device(d_syn[0].data_ptr(), B, d_osdw.data_ptr(), d_osd0.data_ptr(), d_bp.data_ptr(), d_conv.data_ptr(),
                            d_iters.data_ptr(), None)
    dec.synchronize()
    t_last = dec.last_timing()
    # ---- the rows the gather moves, against the byte rows just written: batch 0 once more through the packed device-pointer call
    # (the timed loop's own form where the kernels write packed rows), all three outputs and both per-shot vectors
    packed_same = None
    if do_gather and native_gather:
        c2, i2 = mk(B), mk(B, dtype=torch.int32)
        dec.decode_batch_device_packed(d_psyn[0].data_ptr(), B, d_packed[0].data_ptr(), pouts[0]["osd0"].data_ptr(), pouts[0]["bp"].data_ptr(),
                                       c2.data_ptr(), i2.data_ptr())
        dec.synchronize()
        tmp = torch.empty((B, wpr), dtype=torch.int64, device=dev)
        packed_same = bool(torch.equal(c2, d_conv) and torch.equal(i2, d_iters))
        for byte_rows, words in ((d_osdw, d_packed[0]), (d_osd0, pouts[0]["osd0"]), (d_bp, pouts[0]["bp"])):
            dec.pack_rows_device(byte_rows.data_ptr(), B, n, tmp.data_ptr())
            dec.synchronize()
            packed_same = packed_same and bool(torch.equal(tmp, words))
        del tmp, c2, i2
        if not packed_same:
            print(f"[bench] PACKED ROWS DIFFER FROM THE BYTE ROWS ({P['name']})", file=sys.stderr, flush=True)

    # ---- cross-kernel check (rank 0, outside the timed region): the whole batch once more on the OTHER kernel path, each
    # pinned to the CPU oracle by tests/ -- min-sum: the generic LDS BP kernel and the workgroup OSD kernel (large codes: the
    # any-degree BP kernel); product-sum: the degree-class kernel against the generic LDS kernel (whichever of the two the
    # library did not choose) -- must give the same five outputs bit for bit.  (This is what would have shown the 3-in-131072
    # race of DESIGN.md 4.8 at once.)
    cross = None
    if rank == 0 and a.variant == 0 and not a.no_cross:
        try:
            other = BpOsdDecoder(H, device=local_rank, **kw)
            if large:  # the HBM-resident BP kernel against the any-degree kernel (run-time degree loops); one OSD kernel exists there
                other.set_bp_variant(64)
            elif bp_method == "ps":
                other.set_bp_variant(1 if dec.bp_kernel_info()["kernel"] == "bp_class_kernel" else 32)
                other.set_osd_variant(1)
            else:
                other.set_bp_variant(1)
                other.set_osd_variant(1)
            if other.num_lanes:
                o2 = dict(osdw=mk(B, n), osd0=mk(B, n), bp=mk(B, n), conv=mk(B), iters=mk(B, dtype=torch.int32))
                other.decode_batch_device(d_syn[0].data_ptr(), B, o2["osdw"].data_ptr(), o2["osd0"].data_ptr(), o2["bp"].data_ptr(),
                                          o2["conv"].data_ptr(), o2["iters"].data_ptr(), None)
                other.synchronize()
                same = {k: bool(torch.equal(x, y)) for k, (x, y) in dict(osdw=(d_osdw, o2["osdw"]), osd0=(d_osd0, o2["osd0"]), bp=(d_bp, o2["bp"]),
                                                                         converged=(d_conv, o2["conv"]), iters=(d_iters, o2["iters"])).items()}
                cross = {"identical": all(same.values()), "outputs": same, "shots": int(B),
                         "against": other.bp_kernel_info()["kernel"] + " + " + (other.last_osd_kernel() or "osd_kernel")}
                del o2
            del other
        except Exception as e:  # a second decoder that cannot be built or run leaves the outputs unchecked: not a measurement either
            cross = {"identical": None, "error": f"{type(e).__name__}: {e}"[:200]}
        if cross is not None and cross.get("identical") is False:
            print(f"[bench] CROSS-KERNEL CHECK FAILED ({P['name']}): the two kernel paths disagree on {cross['outputs']}", file=sys.stderr, flush=True)
        if cross is not None and cross.get("identical") is None:
            print(f"[bench] CROSS-KERNEL CHECK DID NOT RUN ({P['name']}): {cross.get('error')}", file=sys.stderr, flush=True)

    # ---- host-to-host leg (rank 0, N = 1): the same batches through the host-pointer API from page-locked buffers
    host = None
    if rank == 0 and world == 1 and a.host_steps > 0 and batches is not None:
        h_syn = [dec.pinned_empty((B, m)) for _ in range(nbatch)]
        for dst, b in zip(h_syn, batches):
            dst[:] = b[1]
        h_out = dict(osdw=dec.pinned_empty((B, n)), osd0=dec.pinned_empty((B, n)), bp=dec.pinned_empty((B, n)),
                     conv=dec.pinned_empty((B,)), iters=dec.pinned_empty((B,), np.int32))
        host = {}
        for label, extra in (("osdw", {}), ("all", dict(osd0=h_out["osd0"], bp=h_out["bp"]))):
            dec.decode_batch_into(h_syn[0], h_out["osdw"], converged=h_out["conv"], iters=h_out["iters"], **extra)  # warm-up
            th = time.perf_counter()
            for k in range(a.host_steps):
                dec.decode_batch_into(h_syn[k % nbatch], h_out["osdw"], converged=h_out["conv"], iters=h_out["iters"], **extra)
            host[label] = (time.perf_counter() - th) / a.host_steps
        # last call decoded batch (host_steps - 1) % nbatch with all outputs: must equal the device-resident result
        if (a.host_steps - 1) % nbatch == 0:
            host["same"] = bool((torch.from_numpy(h_out["osdw"]).to(dev) == d_osdw).all().item() and
                                (torch.from_numpy(h_out["osd0"]).to(dev) == d_osd0).all().item())
        # the same with bit-packed rows across PCIe (bposd_decode_batch_packed): syndromes in, osdw / osd0 / bp out
        wm = (m + 63) // 64
        p_syn = [dec.pinned_empty((B, wm), np.uint64) for _ in range(nbatch)]
        for dst, b in zip(p_syn, batches):
            for lo in range(0, B, 16384):
                dst[lo:lo + 16384] = dec.pack_rows(b[1][lo:lo + 16384])
        p_out = dict(osdw=dec.pinned_empty((B, wpr), np.uint64), osd0=dec.pinned_empty((B, wpr), np.uint64), bp=dec.pinned_empty((B, wpr), np.uint64))
        for label, extra in (("packed_osdw", {}), ("packed_all", dict(osd0_words=p_out["osd0"], bp_words=p_out["bp"]))):
            dec.decode_batch_packed_into(p_syn[0], p_out["osdw"], converged=h_out["conv"], iters=h_out["iters"], **extra)  # warm-up
            th = time.perf_counter()
            for k in range(a.host_steps):
                dec.decode_batch_packed_into(p_syn[k % nbatch], p_out["osdw"], converged=h_out["conv"], iters=h_out["iters"], **extra)
            host[label] = (time.perf_counter() - th) / a.host_steps
        # a STREAM of batches through the asynchronous forms (bposd_decode_batch_async / _packed_async): three calls in flight
        # on three lanes, each with buffers of its own -- what a decoding service does; a lone synchronous call always pays
        # its own upload, its longest-running syndrome and its download
        nsl, ncalls = 3, max(12, 4 * a.host_steps)
        for label, is_packed in (("stream_packed_all", True), ("stream_all", False)):
            if is_packed:
                bufs = [dict(osdw=dec.pinned_empty((B, wpr), np.uint64), osd0=dec.pinned_empty((B, wpr), np.uint64), bp=dec.pinned_empty((B, wpr), np.uint64),
                             conv=dec.pinned_empty((B,)), iters=dec.pinned_empty((B,), np.int32)) for _ in range(nsl)]
                issue = lambda k, b: dec.decode_batch_packed_into(p_syn[k % nbatch], b["osdw"], b["osd0"], b["bp"], b["conv"], b["iters"], wait=False)
            else:
                bufs = [dict(osdw=dec.pinned_empty((B, n)), osd0=dec.pinned_empty((B, n)), bp=dec.pinned_empty((B, n)),
                             conv=dec.pinned_empty((B,)), iters=dec.pinned_empty((B,), np.int32)) for _ in range(nsl)]
                issue = lambda k, b: dec.decode_batch_into(h_syn[k % nbatch], b["osdw"], b["osd0"], b["bp"], b["conv"], b["iters"], wait=False)
            lanes = [None] * nsl
            for k in range(nsl):  # warm-up: buffers of every lane grow here
                lanes[k] = issue(k, bufs[k])
            dec.synchronize()
            th = time.perf_counter()
            for k in range(ncalls):
                sl = k % nsl
                if k >= nsl:
                    dec.synchronize(lanes[sl])
                lanes[sl] = issue(k, bufs[sl])
            dec.synchronize()
            host[label] = (time.perf_counter() - th) / ncalls
            # the last call that decoded batch 0 must equal the synchronous result
            k0 = max(k for k in range(ncalls) if k % nbatch == 0 and k >= ncalls - nsl) if any(k % nbatch == 0 for k in range(ncalls - nsl, ncalls)) else None
            if k0 is not None:
                b = bufs[k0 % nsl]
                ref = p_out if is_packed else h_out
                host[label + "_same"] = bool((b["osdw"] == ref["osdw"]).all() and (b["osd0"] == ref["osd0"]).all() and (b["bp"] == ref["bp"]).all())
            del bufs
        if (a.host_steps - 1) % nbatch == 0:
            dpk = torch.empty((B, wpr), dtype=torch.int64, device=dev)
            same_p = True
            for words, rows in ((p_out["osdw"], d_osdw), (p_out["osd0"], d_osd0), (p_out["bp"], d_bp)):
                dec.pack_rows_device(rows.data_ptr(), B, n, dpk.data_ptr())
                dec.synchronize()
                same_p = same_p and bool((torch.from_numpy(words.view(np.int64)).to(dev) == dpk).all().item())
            host["same_packed"] = same_p
            del dpk

    if rank != 0:
        return None
    steps = max(a.steps, 1)
    value = world * B * steps / elapsed
    # algorithmic bytes of the dominant kernel (BP): SURVEY.md §8(d)
    bytes_per_iter = (4 * E + 2 * n) * 8
    large_note = "algorithmic fp64 message bytes (4E+2n)*8 per executed iteration; "
    large_form = dec.bp_kernel_info()["read_cycles"] if large else 0  # (which form of bp_large_kernel ran: include/bposd_mi355x_debug.h)
    if large and bp_method == "ms":
        # HBM-resident min-sum (round 4): bit->check messages as before (E written, E read); check->bit either as ONE 32-byte
        # record per check (written once, read at least once), or -- per-check data in LDS -- only the second minimum
        # through the workspace (8 bytes per check written, read once by the edge that holds the minimum)
        bytes_per_iter = (2 * E + 2 * n) * 8 + (2 * 8 * m if large_form == 2 else 2 * 32 * m)
        large_note = ("algorithmic bytes (2E+2n)*8 + %s per executed iteration: fp64 bit->check messages through HBM; the check->bit "
                      "messages are rebuilt from two scaled minima per check%s ((4E+2n)*8 with per-edge messages both ways, rounds "
                      "1-3); " % (("2*8*m", ", the first of them and the sign flags resident in LDS, the second through HBM")
                                  if large_form == 2 else ("2*32*m", ", one 32-byte record per check through HBM")))
    large_note += "messages stream through HBM (1.3 MB per syndrome, far beyond LDS)"
    avg_bp_ms = float(np.mean(bp_ms)) if bp_ms else float("nan")
    avg_iters = iters_tot / steps
    algo_bytes = avg_iters * bytes_per_iter + B * (m + n)
    algo_bytes_last = t_last["bp_iterations"] * bytes_per_iter + B * (m + n)
    hbm_algo_gbs = algo_bytes / (avg_bp_ms * 1e-3) / 1e9 if avg_bp_ms > 0 else 0.0
    traffic = osd_traffic = traffic_src = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc)).get(a.config, {})
            traffic = rec.get("hbm_bytes_per_launch")
            osd_traffic = rec.get("osd_hbm_bytes_per_launch")
            traffic_src = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (" +
                           str(rec.get("command", "?")) + "); PMC counters cannot be read from inside the run")
        except Exception:
            traffic = osd_traffic = None
    if a.batch or a.gen == "device":
        traffic = osd_traffic = None  # (the counters were collected at the configuration's own batch)

    # LER of this rank's shard (osdw), definitions of css_decode_sim.py:257-280 for one sector
    ler = ler0 = ler_bp = None
    if code.lz is not None:
        err0 = d_err0 if d_err0 is not None else torch.from_numpy(batches[0][0]).to(dev)
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
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Hd = torch.sparse_csr_tensor(torch.from_numpy(H.indptr.astype(np.int64)), torch.from_numpy(H.indices.astype(np.int64)),
                                     torch.ones(H.nnz, dtype=torch.float32), size=H.shape).to(dev)
    synd_ok = True
    for lo in range(0, B, 8192):
        got = torch.sparse.mm(Hd, d_osdw[lo:lo + 8192].to(torch.float32).T) % 2
        synd_ok = synd_ok and bool((got.T.to(torch.uint8) == d_syn[0][lo:lo + 8192]).all().item())
    conv_frac = float(d_conv.to(torch.float32).mean().item())
    it_cpu = d_iters.cpu().numpy()

    # which BP kernel ran (asked of the library), and what it moves through LDS per syndrome-iteration
    kinfo = dec.bp_kernel_info()
    local_edge = kinfo["kernel"] == "bp_local_kernel"
    code_label = ("[[29524,484]] HGP of a seeded (5,6)-regular 110x132 matrix, hz 14520x29524, " if large else
                  f"{ref_seed[1]} HGP of {ref_seed[0][:-4]} (the reference's example codes), hz {ref_seed[2]}, " if ref400 else
                  "[[1922,50]] HGP (31x31 circulant 1+x^2+x^5) hz 961x1922, ")
    num_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    math_form = None
    if bp_method == "ps":
        math_form = {"ps_math_form": int(a.ps_math_form),
                     "meaning": ("the reference's operation order: tanh(b2c / 2), log of the rounded quotient (1 + x) / (1 - x); four divisions per edge"
                                 if int(a.ps_math_form) == 0 else "two divisions per edge (pm_tanh_half, pm_log_quot)"),
                     "routines": "bp_osd_amd/csrc/portable_math.h (bit-reproducible; the reference calls the platform libm)",
                     "libm_mismatch_shots_of_2048_clip20": 188 if int(a.ps_math_form) == 0 else 207,
                     "libm_mismatch_source": "tests/test_gpu_parity.py::test_config2_product_sum_cs60_vs_golden[clip20] (2048 frozen libm-oracle shots)"}
    hbm_algorithmic = {
        "achieved": hbm_algo_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_algo_gbs / HBM_PEAK_GBS,
        "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_iteration_per_syndrome": bytes_per_iter,
        "frac_isolated": (algo_bytes_last / (t_last["bp_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_last["bp_ms"] > 0 else None,
        "note": (large_note if large else
                 "SURVEY.md 8(d)'s algorithmic fp64 message bytes (4E+2n)*8 per executed iteration over the HBM peak; the messages "
                 "never leave LDS / registers, so this fraction exceeds 1 and is NOT a utilisation"),
    }
    out = {
        "metric": "syndromes decoded/sec (whole node), large HGP 14520x29524 (BASELINE configs[4])" if large else
                  f"syndromes decoded/sec (whole node) + logical error rate, HGP {ref_seed[1]} p=0.05 (reference example)" if ref400 else
                  "syndromes decoded/sec (whole node) + logical error rate, HGP [[1922,50]] p=0.05",
        "value": value,
        "unit": "syndromes/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{a.config}: " + code_label +
                        f"{'min-sum' if bp_method == 'ms' else 'product-sum'} BP"
                        f"{' variable scaling' if bp_method == 'ms' and ms == 0 else (f' scaling {ms}' if bp_method == 'ms' else '')}"
                        f"{f' (ps_clip {ps_clip})' if bp_method == 'ps' and ps_clip else (' (no clipping, as the reference)' if bp_method == 'ps' else '')}, "
                        f"max_iter={max_iter or n}, {osd_method} order {osd_order}, iid bit-flip q={q}",
            "per_gpu_batch": B,
            "global_batch": B * world,
            "sharding": f"independent syndromes, contiguous shards x{world}" +
                        ("" if not do_gather else ", RCCL gather of bit-packed corrections to rank 0" +
                         (" (rows packed by the decode kernels themselves)" if native_gather else " (bposd_pack_rows_device)") +
                         f", {nbuf} packed buffers for {nslots} steps in flight"),
            "bp_variant": a.variant,
            "pipelined_steps": nslots,
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "timed_outputs": ["osdw", "osd0", "bp", "converged", "iters"],
            "batches": (f"{nbatch} distinct seeded batches alternate over the timed steps; the logical error rates, the CPU "
                        "comparison and the isolated kernel times are on batch 0") +
                       ("" if batches is not None else "; drawn on the device (torch generator, seeds 1000 * rank + k), not numpy's PCG64"),
        },
        "logical_error_rate": ler,
        "logical_error_rate_eb": None if ler is None else float(np.sqrt(ler * (1 - ler) / B)),
        "osd0_logical_error_rate": ler0,
        "osd0_logical_error_rate_eb": None if ler0 is None else float(np.sqrt(ler0 * (1 - ler0) / B)),
        "bp_logical_error_rate": ler_bp,
        "bp_logical_error_rate_eb": None if ler_bp is None else float(np.sqrt(ler_bp * (1 - ler_bp) / B)),
        "corrections_reproduce_syndromes": synd_ok,
        "cross_kernel_check": cross,
        "math_form": math_form,
        "bp_converged_fraction": conv_frac,
        "bp_iterations_mean": float(it_cpu.mean()),
        "bp_iterations_p50_p99_max": [float(np.percentile(it_cpu, 50)), float(np.percentile(it_cpu, 99)),
                                      int(it_cpu.max())],
        "osd_invocations_per_step": osd_tot / steps,
        "kernel_ms": {"bp": avg_bp_ms, "osd": float(np.mean(osd_ms)) if osd_ms else 0.0},
        # the same two kernels with nothing else on the GPU (the verification decode after the timed region); inside
        # the timed region consecutive steps overlap, which stretches the per-launch durations above (two launches share the
        # CUs while one drains its stragglers: kernel_ms.bp can exceed ms_per_step)
        "kernel_ms_isolated": {"bp": t_last["bp_ms"], "osd": t_last["osd_ms"]},
        "kernel_only_syndromes_per_s_per_gpu": B / ((t_last["bp_ms"] + t_last["osd_ms"]) * 1e-3),
    }
    pipelined_note = ("; avg_launch_ms is measured inside the timed region, where consecutive steps overlap on two streams "
                      "(rocprofv3 shows the same stretched durations); isolated_launch_ms / frac_isolated are the same kernel "
                      "alone on the GPU" if nslots > 1 else "")
    # ---- "roofline": the dominant kernel against the roofline that BINDS it -- LDS pipe for the LDS-resident min-sum kernels,
    # vector issue for product-sum, HBM for the HBM-resident kernels.  SURVEY.md 8(d)'s HBM-algorithmic figure of the on-chip
    # kernels (message bytes that never leave the CU over the HBM peak: > 1, not a utilisation) is kept under
    # roofline.hbm_algorithmic with its note.
    if large:
        out["roofline"] = {
            "kernel": "bp_large_kernel (BP message passing, messages in HBM)", "bound": "hbm",
            "achieved": hbm_algo_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_algo_gbs / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_iteration_per_syndrome": bytes_per_iter,
            "avg_launch_ms": avg_bp_ms, "isolated_launch_ms": t_last["bp_ms"], "frac_isolated": hbm_algorithmic["frac_isolated"],
            "hbm_algorithmic_frac": hbm_algo_gbs / HBM_PEAK_GBS,
            "note": large_note + pipelined_note,
        }
    elif bp_method == "ms":
        # LDS roofline of the BP kernel: messages that cross LDS per syndrome-iteration, each once in and once out.
        # local-edge kernel: 4 of a check's 6 edges + 2 of a bit's 3 edges = 8m doubles each way (n = 2m);
        # LDS kernel: 6m + 3n = 12m doubles each way.  Peaks per instruction from MI355X_MICROARCH.md.
        doubles = (4 * m + 2 * n) if local_edge else (E + E)
        rd = wr = doubles * 8
        cyc = rd / LDS_READ_B64_BYTES_PER_CLK + wr / LDS_WRITE_B64_BYTES_PER_CLK
        bound_ms = lambda iters: iters * cyc / (num_cu * CLOCK_HZ) * 1e3
        # fp64 VALU issue: per check of degree d  3d - 4 v_min_f64 + d v_mul_f64 + d v_cmp (sign test), per bit of degree
        # d  3d - 3 v_add_f64 + 1 v_cmp (H1922: 26 per check, 7 per bit); a wave64 fp64 instruction issues over 4
        # cycles on one of the CU's 4 SIMDs
        vdeg = np.diff(H.tocsc().indptr)
        fp64_cyc = (float(np.sum(5 * np.diff(H.indptr) - 4)) + float(np.sum(3 * vdeg - 2))) / 64 * 4 / 4
        lds_peak_gbs = (rd + wr) / cyc * num_cu * CLOCK_HZ / 1e9  # bytes per cycle of THIS read / write mix x CUs x clock
        lds_achieved = avg_iters * (rd + wr) / (avg_bp_ms * 1e-3) / 1e9 if avg_bp_ms > 0 else 0.0
        out["roofline_lds"] = {
            "kernel": kinfo["kernel"] + " (BP message passing, LDS- and register-resident messages)",
            "bound": "lds",
            "lds_bytes_per_syndrome_iteration": {"read": rd, "write": wr},
            "lds_cycles_per_syndrome_iteration_per_cu": cyc,
            "peak": {"ds_read_b64_B_per_clk_per_cu": LDS_READ_B64_BYTES_PER_CLK,
                     "ds_write_b64_B_per_clk_per_cu": LDS_WRITE_B64_BYTES_PER_CLK, "cus": num_cu, "clock_hz": CLOCK_HZ},
            "bound_ms_per_launch": bound_ms(avg_iters),
            "frac": bound_ms(avg_iters) / avg_bp_ms if avg_bp_ms > 0 else None,
            "frac_isolated": bound_ms(t_last["bp_iterations"]) / t_last["bp_ms"] if t_last["bp_ms"] > 0 else None,
            "fp64_valu_issue_frac_isolated": (t_last["bp_iterations"] * fp64_cyc / (num_cu * CLOCK_HZ) * 1e3 / t_last["bp_ms"])
            if t_last["bp_ms"] > 0 else None,
            "ns_per_syndrome_iteration_isolated": t_last["bp_ms"] * 1e6 / max(t_last["bp_iterations"], 1),
            "ps_per_edge_iteration_isolated": t_last["bp_ms"] * 1e9 / max(t_last["bp_iterations"], 1) / E,
            "bit_pass_bank_model": {k: kinfo[k] for k in ("read_cycles", "read_floor", "write_cycles", "write_floor")},
            "note": "LDS-pipe cycles the selected kernel needs per syndrome-iteration (conflict-free) x executed iterations / "
                    "(CUs x peak clock) over the measured launch time; fp64_valu_issue_frac is the same ratio for the fp64 "
                    "instruction issue slots",
        }
        out["roofline"] = {
            "kernel": out["roofline_lds"]["kernel"], "bound": "lds",
            "achieved": lds_achieved, "peak": lds_peak_gbs, "unit": "GB/s",
            "frac": lds_achieved / lds_peak_gbs if lds_peak_gbs > 0 else None,
            "frac_isolated": out["roofline_lds"]["frac_isolated"],
            # the same bound over the STEP (ms_per_step: BP, OSD and everything else of a step in steady state, consecutive steps
            # overlapped) -- a launch's own duration always carries its max_iter = n straggler tail (2.3 ms on H1922), which a
            # single persistent launch cannot hide and the next step's launch does
            "frac_of_step": bound_ms(avg_iters) / (1e3 * elapsed / steps) if elapsed > 0 else None,
            "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_ms": avg_bp_ms, "isolated_launch_ms": t_last["bp_ms"],
            "hbm_algorithmic_frac": hbm_algo_gbs / HBM_PEAK_GBS, "hbm_algorithmic": hbm_algorithmic,
            "note": "achieved = message bytes through the LDS pipe per second (read + write, each message once in and once out per "
                    "executed iteration); peak = the same read / write mix at ds_read_b64 256 B/clk/CU and ds_write_b64 85.3 B/clk/CU "
                    "x CUs x peak clock (MI355X_MICROARCH.md); frac is the utilisation of the roofline that binds this kernel "
                    "(= roofline_lds.frac); traffic = measured HBM bytes per launch" + pipelined_note,
        }
    else:
        # VALU roofline of the product-sum BP kernel ("report VALU utilisation too").  Per edge-iteration
        # the check update evaluates one tanh and one log((1+x)/(1-x)) with csrc/portable_math.h; the count of vector
        # instructions per edge-iteration is measured (SQ_INSTS_VALU, profiles/valu_model.json) where a measurement of this
        # evaluation order exists, else the static count of the compiled loop (tools/isa_loop_count.py).
        form_key = a.config + ("" if int(a.ps_math_form) == 0 else "_form1")
        insts = 261.0
        src = "static count of the compiled iteration loop, tools/isa_loop_count.py bp_kernel<6,3,2,4,512,6,true,2,1024>: four divisions per edge"
        vm = os.path.join(ROOT, "profiles", "valu_model.json")
        if os.path.exists(vm):
            try:
                rec = json.load(open(vm)).get(form_key)
                if rec:
                    insts, src = float(rec["valu_wave_insts_per_64_edge_iterations"]), rec["source"]
            except Exception:
                pass
        simd_cycles = lambda iters: iters * E / 64.0 * insts * 4.0  # a wave64 VALU instruction issues over 4 cycles on one SIMD
        bound = lambda iters: simd_cycles(iters) / (num_cu * 4 * CLOCK_HZ) * 1e3
        peak_ginst = num_cu * 4 * CLOCK_HZ / 4.0 / 1e9            # wave-instructions per second the chip can issue
        ach_ginst = avg_iters * E / 64.0 * insts / (avg_bp_ms * 1e-3) / 1e9 if avg_bp_ms > 0 else 0.0
        out["roofline_valu"] = {
            "kernel": kinfo["kernel"], "bound": "valu",
            "valu_wave_insts_per_64_edge_iterations": insts, "source": src,
            "peak": {"simds": num_cu * 4, "clock_hz": CLOCK_HZ, "cycles_per_wave_instruction": 4},
            "bound_ms_per_launch": bound(avg_iters),
            "frac": bound(avg_iters) / avg_bp_ms if avg_bp_ms > 0 else None,
            "frac_isolated": bound(t_last["bp_iterations"]) / t_last["bp_ms"] if t_last["bp_ms"] > 0 else None,
            "ps_per_edge_iteration_isolated": t_last["bp_ms"] * 1e9 / max(t_last["bp_iterations"], 1) / E,
            "note": "vector-instruction issue cycles of the executed edge-iterations over the SIMD-cycles of the launch; the "
                    "kernel is VALU-bound (the messages stay in LDS)",
        }
        out["roofline"] = {
            "kernel": kinfo["kernel"] + " (product-sum BP, messages in LDS)", "bound": "valu",
            "achieved": ach_ginst, "peak": peak_ginst, "unit": "G wave-instructions/s",
            "frac": ach_ginst / peak_ginst, "frac_isolated": out["roofline_valu"]["frac_isolated"],
            "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_ms": avg_bp_ms, "isolated_launch_ms": t_last["bp_ms"],
            "hbm_algorithmic_frac": hbm_algo_gbs / HBM_PEAK_GBS, "hbm_algorithmic": hbm_algorithmic,
            "note": "achieved = vector instructions issued per second by the executed edge-iterations (" + src + "); peak = one wave "
                    "instruction per 4 cycles per SIMD x 4 SIMDs x CUs x peak clock; frac = roofline_valu.frac" + pipelined_note,
        }
    if large:
        # the OSD kernel dominates this configuration; SURVEY.md §8(d) prices it at one read+write pass over the
        # packed matrix plus the sort plus the candidate sweep per invoked syndrome
        W = (n + 1 + 63) // 64
        ncand = (1 << osd_order) - 1 if osd_method == "osd_e" else 0
        osd_bytes = (osd_tot / steps) * (2 * W * 8 * m + n * 12 + ncand * ((m + 63) // 64) * 8)
        avg_osd_ms = float(np.mean(osd_ms)) if osd_ms else float("nan")
        ao = osd_bytes / (avg_osd_ms * 1e-3) / 1e9 if avg_osd_ms > 0 else 0.0
        out["roofline_osd"] = {
            "kernel": "osd_large_kernel (sort + blocked GF(2) elimination + OSD-E sweep, matrix in HBM)",
            "bound": "hbm", "achieved": ao, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ao / HBM_PEAK_GBS,
            "frac_isolated": (osd_bytes / (t_last["osd_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_last["osd_ms"] > 0 else None,
            "traffic": osd_traffic, "algorithmic_bytes_per_launch": osd_bytes, "avg_launch_ms": avg_osd_ms,
            "isolated_launch_ms": t_last["osd_ms"],
            "note": "the single-pass figure of SURVEY.md §8(d); a blocked elimination revisits the trailing matrix once "
                    "per group of pivot panels; since round 5 the trailing update touches only the words that change "
                    "(work list of mask bits, atomic XOR) and its phases are bound by dependent round trips and the vector-memory "
                    "issue rate of scattered 8-byte accesses, not by HBM bandwidth (DESIGN.md §4.5)",
        }
    if per_rank is not None:
        out["per_rank"] = per_rank
    if gather_ms is not None:
        out["gather_ms"] = gather_ms
        out["gather"] = {"backend": ("gloo" if a.rehearse_on_one_gpu else "nccl (RCCL)"), "world_size": world,
                         "native_packed_rows": native_gather, "packed_rows_equal_byte_rows": packed_same, "packed_buffers": nbuf,
                         "launches_that_waited_for_a_gather": pipe.launch_waited_for_gather}
    if a.rehearse_on_one_gpu:
        out["rehearsal"] = True
        out["rehearsal_note"] = ("every rank decoded on device 0 and the gather ran over gloo on host copies of the packed rows: "
                                 "a functional rehearsal of the N > 1 code path, not a measurement")
    if host is not None:
        out["value_host_to_host"] = B / host["osdw"]
        out["host_to_host"] = {
            "unit": "syndromes/s",
            "osdw_converged_iters": B / host["osdw"],
            "all_outputs": B / host["all"],
            "packed_osdw_converged_iters": B / host["packed_osdw"],
            "packed_all_outputs": B / host["packed_all"],
            "stream_all_outputs": B / host["stream_all"],
            "stream_packed_all_outputs": B / host["stream_packed_all"],
            "stream_matches_synchronous_calls": bool(host.get("stream_all_same", True) and host.get("stream_packed_all_same", True)),
            "stream_note": "the same batches as a stream of asynchronous calls (bposd_decode_batch_async / _packed_async), three in flight "
                           "on three lanes with buffers of their own: consecutive calls overlap on the device like the device-resident steps of `value`",
            "ms_per_step": {"osdw_converged_iters": 1e3 * host["osdw"], "all_outputs": 1e3 * host["all"],
                            "packed_osdw_converged_iters": 1e3 * host["packed_osdw"], "packed_all_outputs": 1e3 * host["packed_all"],
                            "stream_all_outputs": 1e3 * host["stream_all"], "stream_packed_all_outputs": 1e3 * host["stream_packed_all"]},
            "steps": a.host_steps,
            "matches_device_resident_run": host.get("same"),
            "packed_matches_device_resident_run": host.get("same_packed"),
            "note": "bposd_decode_batch (numpy in / numpy out; one byte per bit) and bposd_decode_batch_packed (64 bits per word both "
                    "ways, SURVEY.md 8(d)(i) / 8(e)) from page-locked host buffers to page-locked host buffers, PCIe-inclusive, "
                    "chunks overlapped on the handle's lanes; never `value`",
        }
    if cpu is not None:
        got = d_osdw[:cpu["n"]].cpu().numpy()
        rows_same = (got == cpu["osdw"]).all(axis=1) & (it_cpu[:cpu["n"]] == cpu["iters"])
        out["cpu_baseline"] = {
            "value": cpu["n"] / cpu["dt"],
            "unit": "syndromes/s",
            "cores": 1,
            "kind": "port",
            "sample": f"first {cpu['n']} syndromes of batch 0, decoded one at a time by oracle/bposd_oracle.c "
                      f"(single thread, {cpu['dt']:.1f} s); the reference's ldpc/Cython path is not installable here" +
                      ("; product-sum: tanh / log of the platform libm, as the reference calls them" if bp_method == "ps" else ""),
            "host_cores_available": len(os.sched_getaffinity(0)),
            "host_cores_usable": usable_cores(),
            "gpu_matches_cpu_bit_for_bit": bool(rows_same.all()),
        }
        if bp_method == "ps":
            # the kernels evaluate tanh / log with portable_math.h, not with the libm: last-bit differences move a few per cent
            # of the shots -- counted here; the bit-exact statement is against the oracle's portable-math mode
            out["cpu_baseline"]["shots_that_differ_from_the_libm_oracle"] = int((~rows_same).sum())
            out["cpu_baseline"]["shots_compared"] = int(cpu["n"])
            if cpu_pm is not None:
                gp = d_osdw[:cpu_pm["n"]].cpu().numpy()
                out["cpu_baseline"]["gpu_matches_portable_math_oracle_bit_for_bit"] = bool(
                    (gp == cpu_pm["osdw"]).all() and (it_cpu[:cpu_pm["n"]] == cpu_pm["iters"]).all())
                out["cpu_baseline"]["portable_math_oracle"] = (f"first {cpu_pm['n']} syndromes again with the oracle's ps_math = {2 - int(a.ps_math_form)} mode "
                                                               f"(the kernels' routines and evaluation order; {cpu_pm['n'] / cpu_pm['dt']:.0f} syndromes/s): the message "
                                                               "schedule is checked bit for bit, the transcendental code is shared")
        if cpu_all is not None:
            lo, cnt = cpu_all["lo"], cpu_all["procs"] * cpu_all["per"]
            got = d_osdw[lo:lo + cnt].cpu().numpy()
            rows_all = (got == cpu_all["osdw"]).all(axis=1) & (it_cpu[lo:lo + cnt] == cpu_all["iters"])
            out["cpu_baseline_all_cores"] = {
                "value": cnt / cpu_all["wall"],
                "unit": "syndromes/s",
                "cores": cpu_all["procs"],
                "kind": "port",
                "sample": f"syndromes {lo}..{lo + cnt} of batch 0 in {cpu_all['procs']} disjoint shards of {cpu_all['per']}, one "
                          f"forked oracle process per usable host core (wall {cpu_all['wall']:.1f} s, slowest worker "
                          f"{cpu_all['busy']:.1f} s busy)",
                "gpu_matches_cpu_bit_for_bit": bool(rows_all.all()),
            }
            if bp_method == "ps":
                out["cpu_baseline_all_cores"]["shots_that_differ_from_the_libm_oracle"] = int((~rows_all).sum())
    if cross is not None and cross.get("identical") is not True:
        out["value"] = None  # a number whose outputs two implementations do not agree on (or were not compared on) is not a measurement
    if packed_same is False:
        out["value"] = None  # ... nor one whose gathered rows are not the rows the decode wrote
    return out


def summary(rec):
    """What the headline line carries of an extra configuration."""
    r = rec["roofline"]
    return {
        "workload": rec["config"]["workload"], "per_gpu_batch": rec["config"]["per_gpu_batch"], "steps": rec["steps"], "warmup": rec["warmup"],
        "value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"],
        "kernel_ms": rec["kernel_ms"], "kernel_ms_isolated": rec["kernel_ms_isolated"],
        "roofline": {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_isolated", "frac_of_step", "hbm_algorithmic_frac")},
        **({"roofline_osd": {k: rec["roofline_osd"].get(k) for k in ("bound", "frac", "frac_isolated", "avg_launch_ms", "isolated_launch_ms")}}
           if "roofline_osd" in rec else {}),
        "cross_kernel_check": rec["cross_kernel_check"], "math_form": rec["math_form"],
        "logical_error_rate": rec["logical_error_rate"], "bp_converged_fraction": rec["bp_converged_fraction"],
        "bp_iterations_mean": rec["bp_iterations_mean"], "osd_invocations_per_step": rec["osd_invocations_per_step"],
        "corrections_reproduce_syndromes": rec["corrections_reproduce_syndromes"],
        "cpu_baseline": rec.get("cpu_baseline"), "data_batches": rec["config"]["batches"],
    }


def main():
    # A handle's lanes are HIP streams that must be able to run side by side.  The runtime multiplexes a process's streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams that land on one queue run their kernels one after the other;
    # which streams share depends on how many were created before (measured on l29k_ms_e15, three calls in flight: 12.2 k
    # syndromes/s alone in a fresh process, 10.0-10.4 k as the fifth decoder of the default run, 10.9 k with 2 queues, 12.1-12.2 k
    # with 8 in both places).  Read by the runtime when it initialises, i.e. before anything below touches the GPU.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="run this configuration alone (default: h1922_ms_cs7, plus a short run of every other BASELINE configuration "
                         "under \"configs\" when N = 1)")
    ap.add_argument("--no-extras", action="store_true", help="default run without the other configurations")
    ap.add_argument("--extra-steps", type=int, default=4, help="timed steps of each extra configuration of the default run")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--p", type=float, default=0.05, help="bit-flip probability q")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="syndromes timed on one CPU thread (0 = skip both CPU legs; default 8192, 1 for the large code)")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="worker processes of the all-cores CPU leg (0 = skip it; default: usable cores, at most 64)")
    ap.add_argument("--host-steps", type=int, default=3, help="steps of the host-to-host leg (0 = skip)")
    ap.add_argument("--variant", type=int, default=0, help="BP kernel / workgroup shape (0 auto; see bposd_set_bp_variant)")
    ap.add_argument("--ps-math-form", type=int, default=0, choices=(0, 1),
                    help="product-sum evaluation order: 0 the reference's (default), 1 two divisions per edge")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the final RCCL gather")
    ap.add_argument("--force-gather", action="store_true",
                    help="N = 1: initialise the RCCL backend at world size 1 and run the exchange step (packed rows, gather, all-reduce) anyway")
    ap.add_argument("--no-cross", action="store_true", help="skip the cross-kernel check")
    ap.add_argument("--no-pipeline", action="store_true", help="one step at a time (no overlap of consecutive steps)")
    ap.add_argument("--slots", type=int, default=0, help="steps in flight (default 2; more needs that many lanes)")
    ap.add_argument("--max-iter", type=int, default=-1, help="override max_iter (diagnostics; -1 = the config's)")
    ap.add_argument("--gen", default="numpy", choices=("numpy", "device"), help="where the synthetic batch is drawn")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 rehearsal where only one GPU exists: every rank decodes on device 0 and the gather runs over "
                         "gloo on host copies of the packed rows (same StepPipeline, same JSON; not a measurement)")
    args = ap.parse_args()
    args.label = None
    args.nbatch = 2

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    t_start = time.time()
    with_extras = args.config is None and world == 1 and not args.no_extras and not args.force_gather and not args.batch and args.variant == 0
    if args.config is None:
        args.config = "h1922_ms_cs7"
    runs = [args]
    if with_extras:
        for label, cfg, batch, gen in EXTRA_RUNS:
            e = copy.copy(args)
            e.label, e.config, e.batch, e.gen = label, cfg, batch, gen
            e.steps, e.warmup, e.host_steps, e.cpu_procs = args.extra_steps, 1, 0, 0
            if cfg.startswith("l29k"):  # (100 ms steps whose first two still warm the workspaces up: a few more, for a number close to the one of record)
                e.steps, e.warmup = 2 * args.extra_steps, 2
            e.cpu_sample = CPU_SAMPLE_EXTRA.get(cfg, 0) if gen == "numpy" else 0
            # one seeded batch -- two on the large code, like its line of record: a step there costs the SUM of its ~252 eliminations'
            # CU time, and that sum differs by 20-30 % between batches (batch 0 alone: 10.0 k syndromes/s, batches 0 / 1 alternating: 11.9 k)
            e.nbatch = 2 if cfg.startswith("l29k") else 1
            runs.append(e)
    # ---- phase 1: codes, batches and every CPU leg, before the GPU is initialised in this process
    preps = [prepare(r, rank, world) for r in runs]

    import torch
    import torch.distributed as dist

    if not args.rehearse_on_one_gpu and args.gpus > torch.cuda.device_count():
        raise SystemExit(f"--gpus {args.gpus} but this node shows {torch.cuda.device_count()} GPU(s); "
                         "use --rehearse-on-one-gpu for a functional rehearsal of the N > 1 path on one GPU")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_gather:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.force_gather and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    # ---- phase 2: the GPU
    out = measure(preps[0], rank, local_rank, world)
    preps[0]["batches"] = None
    failed = bool(rank == 0 and out is not None and out["value"] is None)
    if rank == 0 and with_extras:
        out["configs"] = {}
        for P in preps[1:]:
            t1 = time.time()
            try:
                rec = measure(P, rank, local_rank, world)
                out["configs"][P["name"]] = summary(rec)
                failed = failed or rec["value"] is None
            except Exception as e:
                out["configs"][P["name"]] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
                failed = True
            out["configs"][P["name"]]["wall_s"] = time.time() - t1
            P["batches"] = None
            torch.cuda.empty_cache()
        out["configs_note"] = ("short runs of the other BASELINE.json configurations in the same process (extra-steps timed steps, one seeded batch -- two alternating on l29k_ms_e15 --, "
                               "cross-kernel check on the whole batch, a one-core CPU sample where it costs < 10 s; their numbers of record "
                               "with both CPU legs: python bench.py --config NAME, kept under profiles/); h1922_ms_cs7_b1048576 = the headline "
                               "configuration with configs[3]'s whole batch of 2^20 syndromes on the one GPU")
        out["wall_s"] = time.time() - t_start
    if world > 1 or args.force_gather:
        # every rank learns of a failed check BEFORE the closing barrier: no rank is left waiting in a collective
        flag = torch.tensor([1.0 if failed else 0.0], device="cpu" if args.rehearse_on_one_gpu else torch.device("cuda", local_rank))
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        failed = bool(flag.item() > 0)
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1 or args.force_gather:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.stdout.flush()
        os._exit(3)


if __name__ == "__main__":
    main()
