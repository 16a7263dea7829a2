#!/usr/bin/env python3
"""bench.py -- eMagLS filter-design throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--concurrent J]

One "step" = one complete filter-set design (both ears, all channels) of BASELINE.json config 3:
getEMagLsFilters, em32 (32 mics, r = 4.2 cm), N = 4, complex SH, 2702 HRIR directions, 512 taps,
48 kHz -- from the angles and HRIRs resident in HBM to the windowed time-domain filters in HBM.
Every rank designs its own filter sets (independent jobs, weak scaling); the only collective is
one RCCL gather of the finished filters to rank 0 inside the timed region.  With --concurrent J
each rank keeps J independent designs (different HRIR sets) in flight on J HIP streams.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the per-bin sweep kernel);
`cpu_baseline` times the NumPy oracle on this host on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def load_inputs(seed_offset=0):
    from emagls_amd import synth
    gpath = os.path.join(ROOT, "tests", "golden", "ref_fixtures.npz")
    if os.path.exists(gpath):
        g = np.load(gpath)
        azi, zen = g["grid/hrirGridAziRad"], g["grid/hrirGridZenRad"]
        maz, mzn = g["grid/micGridAziRad"], g["grid/micGridZenRad"]
    else:
        azi, zen = synth.fibonacci_grid(2702)
        maz, mzn = synth.em32_grid()
    hL, hR = synth.rigid_sphere_hrirs(azi, zen, seed=20250310 + seed_offset)
    return azi, zen, maz, mzn, hL, hR


def parity_check():
    """Secondary metric: deviation of the GPU filters from the oracle by the reference's own rule
    (verifyEMagLs.m:370-395: normalised max abs difference, max spectral dB difference), on a case the oracle
    finishes in seconds (the full-size case is covered by tests/test_gpu_parity.py)."""
    from oracle import emagls_oracle as O
    import emagls_amd as E
    from emagls_amd import synth
    azi, zen = synth.fibonacci_grid(900)
    maz, mzn = synth.em32_grid()
    hL, hR = synth.rigid_sphere_hrirs(azi, zen, taps=64)
    args = (hL, hR, azi, zen, 0.042, maz, mzn, 4, 48000.0, 128, "complex")
    wL, wR = E.getEMagLsFilters(*args)
    oL, oR = O.getEMagLsFilters(*args)
    nd, db, adb = O.assert_all_close_metrics(np.hstack([wL, wR]), np.hstack([oL, oR]))
    rel = float(max(np.abs(wL - oL).max() / np.abs(oL).max(), np.abs(wR - oR).max() / np.abs(oR).max()))
    return {"case": "getEMagLsFilters em32 N=4 complex-SH, 900 dirs, 64-tap HRIRs, 128-tap filters", "rel_complex_error": rel,
            "norm_max_abs_diff": float(nd), "max_abs_db_diff": float(adb), "tolerance": 1e-6}


def cpu_baseline(azi, zen, maz, mzn, hL, hR, nbins_sample=48):
    """Oracle (NumPy restatement of lib/getEMagLsFilters.m) on a bounded sample: everything outside the
    per-bin loop in full, the per-bin loop (lines :85-106) on bins 2..nbins_sample+1 -- which straddle
    k_cut = 43, so both branches are sampled -- scaled to the 512 bins of the workload."""
    from oracle import emagls_oracle as O
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    order, fs, length, r = 4, 48000.0, 512, 0.042

    def run():
        t0 = time.perf_counter()
        nfft, f, P, k_cut = O._design_consts(fs, length, max(1e3, 500 * order))
        smair, simOrder = O.getSMAIRMatrix(order, fs, nfft, r, np.column_stack([maz, mzn]), "complex")
        Yh = O.getSH(simOrder, np.column_stack([azi, zen]), "complex").conj().T
        HL, HR, gL, gR = O._hrir_prologue(hL, hR, nfft, P)
        t1 = time.perf_counter()
        Ps = nbins_sample + 1
        Wl, Wr = O._emagls_core(HL, HR, lambda k: smair[:, :, k - 1] @ Yh, Ps, k_cut, 25)
        t2 = time.perf_counter()
        Wl_full = np.zeros((P, 25), complex)
        Wl_full[:Ps] = Wl
        O._finish(Wl_full, Wl_full, P, nfft, length, False, nfft // 2, nfft // 2 + gR - gL)
        t3 = time.perf_counter()
        return (t1 - t0) + (t3 - t2) + (t2 - t1) * (P - 1) / nbins_sample

    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            t_1 = run()
    else:
        t_1 = run()
    t_all = run()
    ncores = os.cpu_count() or 1
    best, cores = (t_1, 1) if t_1 <= t_all else (t_all, ncores)
    return {"value": 1.0 / best, "unit": "filter sets/s", "cores": cores, "kind": "port",
            "sample": "NumPy oracle (CPU restatement of the MATLAB path, not MATLAB), config 3: SH/modal/HRIR prologue + "
                      "epilogue in full, per-bin SVD loop on %d of 512 bins (2..%d, straddling k_cut=43) scaled x%.2f; "
                      "1 thread %.2f s/set, %d threads %.2f s/set" % (nbins_sample, nbins_sample + 1, 512.0 / nbins_sample,
                                                                   t_1, ncores, t_all)}


def sh_basis_roofline(lib):
    """SH-basis assembly on a launch big enough to be bandwidth bound: D = 2^20 directions, N = 19, real
    basis -> 8*D*S = 3.36 GB written, 16*D bytes read.  Kernel time from HIP events on the
    default stream, inputs and output resident in HBM."""
    import torch
    from emagls_amd import synth
    D, N = 1 << 20, 19
    S = (N + 1) ** 2
    azi, zen = synth.fibonacci_grid(D)
    t_azi = torch.from_numpy(azi).cuda()
    t_zen = torch.from_numpy(zen).cuda()
    out = torch.empty((S, D), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: lib.emagls_sh_basis_device(N, D, C.c_void_p(t_azi.data_ptr()), C.c_void_p(t_zen.data_ptr()), 0,
                                              C.c_void_p(out.data_ptr()), C.c_void_p(st))
    if call() != 0:
        return None
    torch.cuda.synchronize()
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    nbytes = 8.0 * D * S + 16.0 * D
    ach = nbytes / (best * 1e-3) / 1e9
    return {"kernel": "sh_basis_kernel<real>", "bound": "hbm", "dirs": D, "order": N, "algorithmic_bytes": nbytes,
            "ms": best, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}


def plan_batches(steps, concurrent=0, batch=0):
    """(number of batches in flight, designs per batch) for a timed region of `steps` designs.  Explicit --concurrent /
    --batch win; otherwise up to four batches of up to eight designs (a persistent sweep launch covers at most eight),
    sized so that a short timed region wastes no design of a batch."""
    if concurrent > 0 or batch > 0:
        j = max(1, min(concurrent if concurrent > 0 else 32, steps))   # never more in flight than the timed region holds
        bsz = max(1, min(batch if batch > 0 else 8, j, 8))
        return max(1, j // bsz), bsz
    nbatch = max(1, min(4, -(-steps // 8)))
    return nbatch, max(1, min(8, -(-steps // nbatch)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=96)
    ap.add_argument("--concurrent", type=int, default=int(os.environ.get("EMAGLS_BENCH_CONCURRENT", "0")),
                    help="independent designs in flight per GPU (default: up to four batches, sized to the step count)")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("EMAGLS_BENCH_BATCH", "0")),
                    help="designs per batch (<= 8): one persistent sweep launch covers the whole batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sh-roofline", action="store_true")
    args = ap.parse_args()

    import torch  # first: the library then shares torch's HIP runtime (same SONAME libamdhip64.so.7)
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # EMAGLS_BENCH_FORCE_PG=1 runs the collective path (barrier, gather, max-reduce over RCCL) with a process group of one
    # rank too: the N > 1 code is then exercised on a single-GPU box
    use_pg = world > 1 or os.environ.get("EMAGLS_BENCH_FORCE_PG", "0") == "1"
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from emagls_amd import Batch, Plan, _lib as L
    lib = L.load()
    L.check(lib.emagls_set_device(local_rank))
    K, W = args.steps, args.warmup
    nbatch, Bsz = plan_batches(K, args.concurrent, args.batch)
    J = nbatch * Bsz
    W = max(W, 3 * J)   # every batch needs its eager, capturing and first replayed execute before the timed region

    def make_plan(seed_offset, streams=1):
        azi, zen, maz, mzn, hL, hR = load_inputs(seed_offset=seed_offset)
        p = Plan(L.KIND_EMAGLS, os.environ.get("EMAGLS_BENCH_BASIS", "complex"), 4, 48000.0, 512, hL.shape[0], hL.shape[1], 0.042, 32)
        p.set_streams(streams)
        p.set_hrir_grid(azi, zen)
        p.set_mic_grid(maz, mzn)
        p.set_hrirs(hL, hR)
        return p, (azi, zen, maz, mzn, hL, hR)

    # ---- latency of ONE design with nothing else in flight (3 streams: independent branches fork)
    p0, inputs = make_plan(rank * 100, streams=3)
    for _ in range(3):
        p0.execute()
    p0.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        p0.execute()
        p0.synchronize()
    single_ms = (time.perf_counter() - t0) / 8 * 1e3
    # per-stage times and the per-launch duration of the dominant kernel: eager passes with HIP events
    p0.set_streams(1)
    p0.set_profiling(1)
    p0.execute()
    p0.synchronize()
    stages = p0.stage_times()
    p0.set_profiling(2)
    p0.execute()
    p0.synchronize()
    sweep_ms, sweep_n = p0.sweep_kernel_time()
    info = p0.info()
    p0.close()

    # ---- throughput: nbatch batches of Bsz designs each, every batch on its own stream / hardware queue
    plans, batches = [], []
    for b in range(nbatch):
        ps = [make_plan(rank * 100 + b * Bsz + j)[0] for j in range(Bsz)]
        plans.append(ps)
        batches.append(Batch(ps) if Bsz > 1 else None)
    out = torch.zeros((K, 2, info.out_cols, info.out_rows, 2), dtype=torch.float64, device="cuda")  # complex as (re,im)
    gathered = [torch.zeros_like(out) for _ in range(world)] if (use_pg and rank == 0) else None

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(nsteps, store):
        s = 0
        while s < nsteps:
            started = []
            for b in range(nbatch):
                if s + len(started) * Bsz >= nsteps:
                    break
                if batches[b] is not None:
                    batches[b].execute()
                else:
                    plans[b][0].execute()
                started.append(b)
            for b in started:
                if batches[b] is not None and store and s + Bsz <= nsteps:
                    # one synchronisation and one status check for the whole batch; device-to-device copies
                    batches[b].get_filters_into([out[s + j, 0].data_ptr() for j in range(Bsz)],
                                                [out[s + j, 1].data_ptr() for j in range(Bsz)])
                    s += Bsz
                    continue
                for j, p in enumerate(plans[b]):
                    if s < nsteps and store:
                        L.check(lib.emagls_plan_get_filters(p._h, C.c_void_p(out[s, 0].data_ptr()),
                                                            C.c_void_p(out[s, 1].data_ptr())))
                    elif batches[b] is not None:
                        batches[b].synchronize()
                    else:
                        p.synchronize()
                    s += 1

    run_steps(W, False)  # first execute is eager, the second captures the hipGraph, the third replays it
    if use_pg:  # warm the collective too
        dist.gather(out, gathered, dst=0)
    # ---- timed region: K designs + one gather (hipGraph replays; two HIP events bracket each batch's sweep launch)
    for b in batches:
        if b is not None:
            b.set_profiling(1)
    barrier()
    t0 = time.perf_counter()
    run_steps(K, True)
    if use_pg:
        dist.gather(out, gathered, dst=0)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if use_pg:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # duration of the dominant kernel's launches inside the timed region (the last execute of every batch)
    batch_sweep_ms = [b.sweep_time_ms() for b in batches if b is not None]

    if rank == 0:
        D, Cc = inputs[4].shape[1], info.num_channels
        # algorithmic bytes of one sweep launch (one frequency bin, both ears): the bin's pwGrid (D x C complex),
        # its C x C matrix M_k, |H| of both ears, W(k-1) in and W(k) out
        bytes_bin = 16.0 * D * Cc + 16.0 * Cc * Cc + 2 * 8.0 * D + 2 * 2 * Cc * 16.0
        nbins_swept = info.num_pos_freqs - max(info.k_cut - 1, 1)
        roof = None
        if sweep_n > 0:
            persistent = sweep_n == 1  # one resident launch walks all swept bins (sweep_persist.hip)
            kname = "sweep_persist_kernel" if persistent else "sweep_half_kernel"
            bytes_launch = bytes_bin * nbins_swept / sweep_n
            traffic = None
            try:  # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc cannot run inside the bench)
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    traffic = json.load(f)[kname]["bytes"]
            except Exception:
                traffic = None
            # one design per launch: sweep stage time of the single-design plan (HIP events on the plan's stream)
            stage_sweep_ms = dict(stages).get("magls_sweep", sweep_ms)
            single_s = stage_sweep_ms / sweep_n * 1e-3
            designs_per_launch = 1
            avg_s = single_s
            if persistent and batch_sweep_ms and Bsz > 1:
                # the timed region launches the kernel once per batch of Bsz designs (design j on XCD j)
                designs_per_launch = Bsz
                avg_s = sum(batch_sweep_ms) / len(batch_sweep_ms) * 1e-3
                bytes_launch *= Bsz
                traffic = traffic * Bsz if traffic is not None else None
            ach = bytes_launch / avg_s / 1e9
            roof = {"kernel": kname, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "launches_per_step": sweep_n / designs_per_launch,
                    "designs_per_launch": designs_per_launch, "avg_launch_us": avg_s * 1e6,
                    "avg_launch_us_single_design": single_s * 1e6,
                    "algorithmic_bytes_per_launch": bytes_launch, "bins_per_launch": nbins_swept / sweep_n,
                    "us_per_bin": avg_s * 1e6 * sweep_n / nbins_swept,
                    "note": "sequential recurrence over the frequency bins (W(k) needs W(k-1)): 1.1 MB of operands per bin and "
                            "design; one launch sweeps the designs of a batch, each on its own XCD; the chain is bound by the "
                            "per-bin exchange of partial sums between workgroups (two in-launch hops through the XCD's L2) and "
                            "LDS-bound phases, not by HBM bandwidth -- see DESIGN.md section 5"}
        # the reference formulation needs F_ref FP64 flop per set (pwGrid GEMM + SVD-equivalent + apply);
        # the factorised pipeline executes F_exec (per stage in DESIGN.md section 5)
        Kb, Sx = info.num_pos_freqs - 1, info.num_sh_sim
        f_ref = (8.0 * Kb * Cc * Sx * D + 8.0 * Kb * D * Cc * Cc + 8.0 * 2 * Kb * 2 * D * Cc + 8.0 * Kb * D * Cc * Cc) / 1e9
        # (the SH machinery of a complex-basis design runs in real arithmetic: real Gram, Cholesky and order terms at 2 flop
        # per multiply-add, complex-times-real products at 4, complex-times-complex at 8)
        f_exec = (2.0 * Sx * Sx * D + 2.0 * Sx ** 3 / 3 + 2.0 * D * Cc * Sx + 80.0 * nbins_swept * Cc * D
                  + 4.0 * 2 * (info.k_cut - 1) * D * Sx + 4.0 * Kb * Sx * Cc * 10 + 8.0 * nbins_swept * Sx * Cc * (Cc + 1) / 2
                  + 8.0 * nbins_swept * 4 * D * Cc + 5.0 * D * info.nfft * 10) / 1e9
        flops = {"F_ref_gflop_per_set": f_ref, "F_exec_gflop_per_set": f_exec, "exec_tflops": f_exec * world * K / dt / 1e3,
                 "ref_equivalent_tflops": f_ref * world * K / dt / 1e3, "fp64_peak_tflops": 78.6,
                 "note": "F_ref: reference formulation (SURVEY 8d: 126 GFLOP at config 3); F_exec: flops the factorised pipeline "
                         "executes (Gram on MFMA counted as full tiles); fp64 peak = AMD's public vector/matrix figure"}
        res = {
            "metric": "eMagLS filter sets/s (N=4, 2702 dirs, 512 taps)", "value": world * K / dt, "unit": "filter sets/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: getEMagLsFilters em32 r=4.2cm N=4 complex-SH, 2702 dirs, 512 taps, "
                                   "48 kHz; one filter set per step, inputs resident in HBM",
                       "dirs": int(D), "taps": 512, "sim_order": info.sim_order, "bins": info.num_pos_freqs - 1,
                       "k_cut": info.k_cut, "designs_in_flight_per_gpu": J, "designs_per_batch": Bsz, "batches_in_flight": nbatch,
                       "parallelism": "independent jobs per GPU, one RCCL gather"},
            "roofline": roof,
            "flops": flops,
            "single_design_latency_ms": round(single_ms, 4),
            "stages_ms": {k: round(v, 4) for k, v in stages},
        }
        if not args.no_sh_roofline and world == 1:
            try:
                res["sh_basis_roofline"] = sh_basis_roofline(lib)
            except Exception as e:  # the large launch needs ~3.5 GB; never fail the bench on it
                res["sh_basis_roofline"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(*inputs)
            res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
            res["parity"] = parity_check()
        print(json.dumps(res))
    for b in batches:
        if b is not None:
            b.close()
    for ps in plans:
        for p in ps:
            p.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
