#!/usr/bin/env python3
"""Benchmark of the `autocycler compress` hot path on MI355X.

Metric (BASELINE.json): Mbp/s through compress -> unitig graph (k=51), workload config C = 96 x ~5 Mbp synthetic
assemblies on one GPU.  A "step" is one full pass of the replaced region (compress.rs:42-44).  Three brackets are timed in
the same run and printed in the same JSON line; at N = 1:

  value / ms_per_step   T_hot of SURVEY.md 8(d), the bracket the metric and the >= 50x target are defined on: the padded,
                        end-repaired sequences in (pageable) HOST RAM -> final unitig graph (segments, links in L-line order,
                        paths) in host RAM through `ac_compress_build` — what the Rust shim at compress.rs:42-44 calls —
                        i.e. including the host-side 2-bit pack, the pinned-ring upload (H2D) and the D2H of the results
                        (`t_hot` repeats the bracket with its upload figures)
  hbm_resident          the same region with the text already RESIDENT IN HBM (`ac_compress_build_device`): device build +
                        D2H, same number of steps, same barriers — an extra key, never `value` (rounds 1-3 quoted it as
                        `value`; N > 1 still does, the torch driver keeps the ranks' texts on their devices)
  t_e2e                 the whole command (compress.rs:34 -> :49): FASTA directory -> input_assemblies.gfa / .yaml through
                        `ac_compress_dir` in this (warm) process, and through the `autocycler-compress` CLI in a fresh process
                        (HIP context creation and code-object load included)
plus cold_first_build_ms, the first build of this process (arena, pinned pools, code objects).  Input generation happens
before all of them.

    python bench.py [--gpus N --steps K --warmup W] [--assemblies 96 --genome 5000000 --kmer 51]

N > 1: launched by torch.distributed.run, one rank per GPU.  Default `--mode sharded`: ONE compress job of
N x 96 assemblies, sequences sharded by rank (rank r holds assemblies 96r .. 96r+95), the job's k-mer table partitioned over the
ranks by key hash; the ranks exchange their novel-run fragments (all-gather), the owners' bitmap / degree / link contributions (SUM
all-reduce), the walk-start keys and the per-unitig depths / positions over RCCL — autocycler_amd/sharded.py.  Weak scaling: 96 assemblies per GPU; by default the job is a mixed-species one (species r
on rank r, like BASELINE.json configs[4]) so that the output per GPU is fixed too; `--species one` makes all N x 96
assemblies one species (the P lines of such a job grow with N^2).
`--mode independent`: every rank builds the graph of its own 96-assembly set (N unrelated jobs, no data-path collective).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

A_K = lambda k: 1 + 2 * (8 * ((2 * k + 63) // 64) + 8)   # algorithmic bytes per input bp
HBM_PEAK = 8.0e12


def workload_label(args, k):
    """Which BASELINE.json configuration (if any) the arguments describe."""
    key = (args.assemblies, args.genome, k)
    std_noise = (args.sub, args.indel) == (1e-4, 1e-5)
    if std_noise and args.plasmid == 100_000 and key == (96, 5_000_000, 51):
        return "BASELINE.json configs[2]"
    if std_noise and args.plasmid == 100_000 and key == (12, 5_000_000, 51):
        return "BASELINE.json configs[1]"
    if std_noise and key == (24, 100_000_000, 101):
        return "BASELINE.json configs[3] on ONE GPU"
    if std_noise and key == (24, 10_000_000, 101):
        return "D' (scaled replica of BASELINE.json configs[3])"
    return "custom workload (no BASELINE.json configuration)"


def make_inputs(args, rank, sharded_job):
    """sharded_job: this rank's slice (assemblies rank*A .. rank*A + A-1) of ONE job; else an unrelated job per rank.
    A sharded job is by default a mixed-species one (BASELINE.json configs[4] is): rank r holds the A assemblies of
    species r, so the work AND the output per GPU stay fixed as N grows (weak scaling).  `--species one`: all N x A
    assemblies are of one species — every sequence then runs through N times as many unitigs, i.e. the P lines of the
    job grow with N^2 (that is the reference's output for such a job, not overhead; it is not a fixed per-GPU work)."""
    import numpy as np
    from autocycler_amd import synth
    first = rank * args.assemblies if sharded_job else 0
    if sharded_job and args.species == "one":
        asm = synth.make_assemblies(args.assemblies, genome=args.genome, plasmid=args.plasmid, sub=args.sub, indel=args.indel,
                                    seed=51_000, first=first)
    else:      # species `rank` (rank 0: exactly the N = 1 workload)
        asm = synth.make_assemblies(args.assemblies, genome=args.genome, plasmid=args.plasmid, sub=args.sub, indel=args.indel,
                                    seed=51_000 + 1000 * rank)
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(asm):
        for header, s in contigs:
            seqs.append(np.ascontiguousarray(s)); fn.append(f"assembly_{first + i:04d}.fasta"); hd.append(header)
    return seqs, fn, hd


def prepare(lib, k, seqs, fn, hd, n_assemblies, threads, repair=1):
    """Sequence::new_with_seq (+ sequence_end_repair when repair=1: ac_seqs_from_raw runs it on the device text and brings the ends back) — upstream of the timed region."""
    from autocycler_amd import _capi
    n = len(seqs)
    ptrs = (C.c_void_p * n)(*[s.ctypes.data for s in seqs])
    lens = (C.c_uint32 * n)(*[len(s) for s in seqs])
    f = (C.c_char_p * n)(*[x.encode() for x in fn])
    h = (C.c_char_p * n)(*[x.encode() for x in hd])
    out = C.c_void_p()
    rc = lib.ac_seqs_from_raw(C.c_uint32(k), C.c_uint32(n), ptrs, lens, f, h, C.c_uint32(n_assemblies), C.c_int(repair),
                              C.c_int(threads), C.byref(out))
    if rc:
        raise RuntimeError(lib.ac_last_error().decode())
    return out


def cpu_baseline(k, sample_assemblies, sample_genome):
    """The oracle (C++ restatement of the reference CPU path, hot stages on 1 core like the reference) timed on
    a bounded sample of the same workload model."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as O
    from autocycler_amd import synth
    asm = synth.make_assemblies(sample_assemblies, genome=sample_genome, plasmid=sample_genome // 50, seed=51_000)
    seqs, fn, hd = [], [], []
    for i, contigs in enumerate(asm):
        for header, s in contigs:
            seqs.append(s.tobytes().decode()); fn.append(f"assembly_{i:04d}.fasta"); hd.append(header)
    s = O.Seqs.from_raw(k, seqs, filenames=fn, headers=hd, repair=True, threads=os.cpu_count() or 1)
    _, _, tm = s.compress(k)
    hot = tm["kmer_graph"] + tm["unitig_graph"] + tm["simplify"]
    bases = sum(len(x) for x in seqs)
    return {"value": bases / 1e6 / hot, "unit": "Mbp/s", "cores": 1, "kind": "port",
            "sample": f"{sample_assemblies} x {sample_genome} bp synthetic assemblies (same generator), k={k}: "
                      f"k-mer graph {tm['kmer_graph']:.2f}s + unitig graph {tm['unitig_graph']:.2f}s + simplify {tm['simplify']:.2f}s "
                      f"on 1 core (the reference's hot stages are single-threaded); C++ restatement, not the autocycler binary"}


def collect_pmc_live(n_text, tag="bench_live", timeout=420, workload=None):
    """HBM-side bytes per kernel per build, counted now: the two rocprofv3 PMC passes of tools/pmc_lean.sh (FETCH_SIZE, WRITE_SIZE;
    --kernel-trace only, one counter per pass) on the torch-free driver building THIS workload with THIS library, summarised by
    tools/pmc_traffic.py.  Returns the dictionary profiles/pmc_traffic.json holds, or an {"error": ...} one."""
    import shutil
    import subprocess
    if not shutil.which("rocprofv3"):
        return {"error": "rocprofv3 not on PATH"}
    try:
        pr = subprocess.run(["bash", str(ROOT / "tools" / "pmc_lean.sh"), tag, "base"] + ([workload] if workload else []), cwd=str(ROOT), capture_output=True, text=True, timeout=timeout,
                            env={**os.environ, "AC_NO_TORCH": "1"})
        out = ROOT / "gpurun_out"
        f, w = out / f"{tag}_pmc_FETCH_SIZE.csv", out / f"{tag}_pmc_WRITE_SIZE.csv"
        if not (f.exists() and w.exists() and f.stat().st_size and w.stat().st_size):
            return {"error": "the PMC passes left no summaries: " + (pr.stdout + pr.stderr)[-300:]}
        pr2 = subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_traffic.py"), str(f), str(w), str(n_text), tag, "7", workload or "configC_k51"], cwd=str(ROOT),
                             capture_output=True, text=True, timeout=60)
        if pr2.returncode:
            return {"error": pr2.stderr[-300:]}
        return json.loads(pr2.stdout)
    except Exception as e:      # noqa: BLE001 — a failed collection leaves `traffic` null, it does not take the line with it
        return {"error": repr(e)[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--assemblies", type=int, default=96)
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--plasmid", type=int, default=100_000)
    ap.add_argument("--sub", type=float, default=1e-4)
    ap.add_argument("--indel", type=float, default=1e-5)
    ap.add_argument("--kmer", type=int, default=51)
    ap.add_argument("--workload", type=str, default=None,
                    help="N = 1: a named workload of autocycler_amd.synth.WORKLOADS instead of --assemblies / --genome / --kmer, e.g. configEprime_k51 "
                         "(the mixed-species replica of BASELINE.json configs[4]); its PMC traffic is profiles/pmc_traffic_<name>.json")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-bracket", action="store_true", help="skip the T_hot bracket (host RAM -> host RAM through ac_compress_build)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the whole-command bracket (FASTA directory -> GFA/YAML)")
    ap.add_argument("--cpu-sample", type=str, default="4x1000000")
    ap.add_argument("--pmc", choices=["auto", "live", "file", "off"], default="auto",
                    help="where roofline.traffic comes from: `file` = profiles/pmc_traffic.json if it was collected on exactly this library's "
                         "sources (else null), `live` = two rocprofv3 PMC passes now (tools/pmc_lean.sh, ~1 min), `auto` = the file when its source "
                         "hash matches, else live")
    ap.add_argument("--mode", choices=["auto", "single", "sharded", "independent"], default="auto",
                    help="auto: single-device build at N=1, one sharded job at N>1")
    ap.add_argument("--init-builds", type=int, default=2, help="untimed builds before the warmup (one-time process initialisation)")
    ap.add_argument("--init-seconds", type=float, default=1.5, help="keep running untimed builds until this much time has passed (device clocks ramp up under load)")
    ap.add_argument("--repair", choices=["device", "host"], default="device",
                    help="where the sequences are when sequence_end_repair runs (upstream of the timed region; the device kernels either way): 'device' = "
                         "on the resident device text (ac_end_repair_device), 'host' = through ac_seqs_from_raw (host sequences up, repaired ends back)")
    ap.add_argument("--no-independent", action="store_true", help="N > 1: skip the secondary independent-jobs measurement")
    ap.add_argument("--no-multi-entry", action="store_true", help="N > 1: skip the secondary measurement of the same job through ac_compress_build_multi (one process, N devices)")
    ap.add_argument("--species", choices=["per-gpu", "one"], default="per-gpu",
                    help="sharded mode at N > 1: one species per GPU (mixed-species job, fixed work and output per GPU) or all "
                         "N x A assemblies of one species (path output grows with N^2)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None,
                    help="N > 1: torch.distributed backend of the sharded job. nccl = RCCL over xGMI (default); gloo = the collectives staged through host "
                         "memory — the verified fallback should inter-device RCCL misbehave on a box (no inter-device transfer has ever run: DESIGN.md 7)")
    ap.add_argument("--protocol-always", action="store_true",
                    help="sharded mode at N = 1: run every phase of the N-rank protocol (fragments, union text, second insert, ...) instead of handing "
                         "the job to the single-device build — what the protocol itself costs before a byte moves")
    ap.add_argument("--gather-paths", action="store_true",
                    help="sharded mode: gather the paths of all sequences to rank 0 (default: every rank keeps its own P lines)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from autocycler_amd import _capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Dry runs of the N > 1 code path on a box with fewer GPUs than ranks: BENCH_FORCE_DEVICE=0 puts every rank on device 0 and
    # BENCH_BACKEND=gloo moves the collectives through the host (RCCL refuses two ranks on one device).
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    backend = args.backend or os.environ.get("BENCH_BACKEND", "nccl")
    # BENCH_EMU_LIB=<path of tests/_emu/libautocycler_emu.so>: dry run of THIS SCRIPT's plumbing (rank layout, collectives over gloo,
    # the JSON line) on the CPU emulation of the kernels — used by tests/test_host_side.py only; its numbers mean nothing and the
    # line says so ("data": "emulation dry run").  Without it the product library is loaded and a GPU is required.
    emu_lib = os.environ.get("BENCH_EMU_LIB")
    if emu_lib:
        backend = "gloo"
        dev = torch.device("cpu")
        device_sync = lambda: None
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        device_sync = torch.cuda.synchronize
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    named = None
    if args.workload:
        from autocycler_amd import synth as _synth
        if world != 1:
            raise SystemExit("--workload names a single-device job")
        if args.workload not in _synth.WORKLOADS:
            raise SystemExit(f"--workload: one of {sorted(_synth.WORKLOADS)}")
        named = _synth.WORKLOADS[args.workload]
        args.kmer, args.assemblies = named[0], named[1]
    lib = _capi.load_library(emu_lib) if emu_lib else _capi.load_library()          # raises if the HIP extension is missing: no fallback
    lib.ac_seqs_views.restype = C.POINTER(_capi.SeqView)
    lib.ac_seqs_count.restype = C.c_uint32
    lib.ac_seqs_free.argtypes = [C.c_void_p]
    lib.ac_seqs_repair_seconds.restype = C.c_double
    k = args.kmer

    mode = args.mode
    if mode == "auto":
        mode = "single" if world == 1 else "sharded"
    if mode == "single" and world > 1:
        mode = "independent"
    t0 = time.time()
    if named:
        from autocycler_amd import synth as _synth
        seqs, fn, hd = _synth.flatten(named[2]())
    else:
        seqs, fn, hd = make_inputs(args, rank, mode == "sharded")
    t_gen = time.time() - t0
    lib.ac_set_host_side_device(C.c_int(local_rank))      # (one process per GPU: the host-side helpers' end repair runs on this rank's device)
    h_seqs = prepare(lib, k, seqs, fn, hd, args.assemblies, threads=os.cpu_count() or 1, repair=1 if args.repair == "host" else 0)
    del seqs
    n = lib.ac_seqs_count(h_seqs)
    views = lib.ac_seqs_views(h_seqs)
    n_text = lib.ac_text_size(C.c_uint32(k), views, C.c_uint32(n))
    import numpy as np
    text = np.empty(n_text, dtype=np.uint8)
    off = (C.c_uint64 * n)(); d1 = (C.c_uint16 * n)(); d2 = (C.c_uint16 * n)()
    if lib.ac_layout_text(C.c_uint32(k), views, C.c_uint32(n), text.ctypes.data_as(C.c_void_p), off, d1, d2):
        raise RuntimeError(lib.ac_last_error().decode())
    lens = (C.c_uint32 * n)(*[views[i].length for i in range(n)])
    ids = (C.c_uint16 * n)(*[views[i].id for i in range(n)])
    bases = sum(lens)
    t_repair = lib.ac_seqs_repair_seconds(h_seqs)
    t1 = time.time()
    d_text = torch.from_numpy(text).to(dev)      # inputs resident in HBM before the timed region
    device_sync()
    t_h2d = time.time() - t1
    repair_info = {"where": args.repair, "seconds": t_repair}
    if args.repair == "device":      # sequence_end_repair on the device text, in place (upstream of the timed region)
        secs, nm = C.c_double(), C.c_uint64()
        t2 = time.time()
        if lib.ac_end_repair_device(C.c_uint32(k), C.c_void_p(d_text.data_ptr()), C.c_uint64(n_text), off, lens, d1, d2, C.c_uint32(n),
                                    C.c_int(local_rank), C.byref(secs), C.byref(nm)):
            raise RuntimeError(lib.ac_last_error().decode())
        lib.ac_end_repair_device(C.c_uint32(k), C.c_void_p(d_text.data_ptr()), C.c_uint64(n_text), off, lens, d1, d2, C.c_uint32(n),
                                 C.c_int(local_rank), C.byref(secs), C.byref(nm))      # idempotent on a repaired text: steady-state time
        repair_info = {"where": "device", "first_call_s": time.time() - t2 - secs.value, "seconds": secs.value, "matches": nm.value}
        t_repair = secs.value

    shard = None
    if mode == "sharded":
        from autocycler_amd import sharded
        id0 = 0
        if world > 1:     # job-wide sequence ids: rank order = sequence order
            counts = torch.zeros(world, dtype=torch.int64, device=dev)
            counts[rank] = n
            dist.all_reduce(counts)
            id0 = int(counts[:rank].sum().item())
        shard = sharded.LocalShard(k, args.assemblies, d_text, n_text, list(off), list(lens), [id0 + i + 1 for i in range(n)],
                                   list(d1), list(d2))
    last_info = {}

    def step():
        if shard is not None:
            g, info = sharded.sharded_build(lib, shard, sharded.Comm(dev), device_index=local_rank, root=0,
                                            gather_paths=args.gather_paths, direct_when_alone=not args.protocol_always)
            last_info.update(info)
            return g
        h = C.c_void_p()
        rc = lib.ac_compress_build_device(C.c_uint32(k), C.c_uint32(args.assemblies), C.c_void_p(d_text.data_ptr()),
                                          C.c_uint64(n_text), off, lens, ids, d1, d2, C.c_uint32(n), C.c_int(local_rank), C.byref(h))
        if rc:
            raise RuntimeError(lib.ac_last_error().decode())
        return _capi.Graph(lib, h, n)

    def barrier():
        device_sync()
        if world > 1:
            dist.barrier()
        device_sync()

    # One-time initialisation outside both the warmup and the timed region: the first two builds of a process load
    # the code objects, create the pinned result pool and the copy stream (60-180 ms and ~20 ms instead of ~9 ms).
    cold_first_build_ms = None
    t_init = time.perf_counter()
    i = 0
    # (... and the device's clocks: a process's first 0.3-1.3 s of builds run ~19 % slower than its later ones — E' 23.1 vs 19.4 ms,
    # mini-E 89 vs 73 ms, whichever library version, with or without per-stage synchronisation: the power management ramping up, DESIGN.md
    # section 6 — so the untimed builds go on until --init-seconds of them have run)
    # (N > 1: every rank must run the same number of builds — they are collective — so a fixed count there)
    n_fixed = args.init_builds if world == 1 else (max(args.init_builds, 8) if args.init_seconds > 0 else args.init_builds)
    while i < n_fixed or (world == 1 and time.perf_counter() - t_init < args.init_seconds and i < 400):
        tc = time.perf_counter()
        step().close()
        if i == 0:
            cold_first_build_ms = (time.perf_counter() - tc) * 1e3
        i += 1
    init_builds_run = i
    if os.environ.get("BENCH_STAGE_TIMING"):      # experiment: keep the per-stage syncs inside the timed loop
        lib.ac_set_stage_timing(C.c_int(1))
    for _ in range(args.warmup):
        step().close()
    barrier()
    t_start = time.perf_counter()
    tms = []
    g = None
    step_s = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        if g is not None:
            g.close()
        g = step()
        step_s.append(time.perf_counter() - ts)      # the build call returns with the graph in host RAM
        tms.append(g.timings())
    barrier()
    elapsed = time.perf_counter() - t_start
    # stage breakdown from two extra, untimed, instrumented builds (a stream sync per stage costs ~0.3 ms per build)
    lib.ac_set_stage_timing(C.c_int(1))
    stage_tms = []
    for _ in range(2):
        g.close()
        g = step()
        stage_tms.append(g.timings())
    lib.ac_set_stage_timing(C.c_int(0))
    barrier()

    # ---- T_hot: the same sequences in host RAM -> final graph in host RAM through ac_compress_build ------------
    t_hot = None
    tms_hot = []
    if world == 1 and mode == "single" and not args.no_host_bracket:
        import hashlib
        # what the Rust caller holds at compress.rs:41: the padded, end-repaired forward sequences in ordinary (pageable) host
        # memory — here the repaired device text read back, one view per sequence
        text_host = d_text.cpu().numpy() if not emu_lib else text
        hviews = (_capi.SeqView * n)()
        base_addr = text_host.ctypes.data
        for i in range(n):
            hviews[i].fwd = C.cast(C.c_void_p(base_addr + off[i]), C.c_char_p)
            hviews[i].length = lens[i]
            hviews[i].id = ids[i]

        def step_host():
            h = C.c_void_p()
            if lib.ac_compress_build(C.c_uint32(k), C.c_uint32(args.assemblies), hviews, C.c_uint32(n), C.c_int(local_rank), C.byref(h)):
                raise RuntimeError(lib.ac_last_error().decode())
            return _capi.Graph(lib, h, n)

        md5_dev = hashlib.md5(g.gfa(fn, hd).encode()).hexdigest()
        tc = time.perf_counter()
        gh = step_host()
        first_host_ms = (time.perf_counter() - tc) * 1e3      # allocates the pinned staging ring
        md5_host = hashlib.md5(gh.gfa(fn, hd).encode()).hexdigest()
        if md5_host != md5_dev:
            raise SystemExit(f"host entry and device entry built different graphs: {md5_host} != {md5_dev}")
        gh.close()
        for _ in range(max(args.warmup, 1)):
            step_host().close()
        barrier()
        th0 = time.perf_counter()
        hs, hup, hupd, tms_hot = [], [], [], []
        gh = None
        for _ in range(args.steps):
            ts = time.perf_counter()
            if gh is not None:
                gh.close()
            gh = step_host()
            hs.append(time.perf_counter() - ts)
            tms_hot.append(gh.timings())
            hup.append(tms_hot[-1]["h2d"]); hupd.append(tms_hot[-1]["upload_device_ms"])
        barrier()
        e_hot = time.perf_counter() - th0
        gh.close()
        t_hot = {"value": bases / 1e6 / (e_hot / args.steps), "unit": "Mbp/s", "ms_per_step": e_hot / args.steps * 1e3, "steps": args.steps,
                 "elapsed_s": e_hot, "step_ms_list": [round(x * 1e3, 2) for x in hs],
                 "timed_region": "padded+repaired sequences in pageable host RAM (one buffer per sequence view) -> final unitig graph in host "
                                 "RAM through ac_compress_build: 2-bit pack on the host (AC_UPLOAD_THREADS = 24 background threads by default) straight into device memory where the "
                                 "device's memory is host-visible (large BAR; else into a pinned ring sent as 16 MB copies): 0.25 B per base over PCIe, "
                                 "the mask plane derived on the device, the insert issued piece by piece as the 64 MB chunks land, device build, D2H "
                                 "of the results",
                 "step_ms": {"min": min(hs) * 1e3, "median": sorted(hs)[len(hs) // 2] * 1e3, "max": max(hs) * 1e3},
                 "upload_ms": sum(hupd) / len(hupd), "upload_GBps": (n_text / (sum(hupd) / len(hupd) * 1e-3) / 1e9) if sum(hupd) else None,      # text bytes per second
                 "upload_pcie_GBps": (0.25 * n_text / (sum(hupd) / len(hupd) * 1e-3) / 1e9) if sum(hupd) else None,      # what crosses the link: 2 bits per base
                 "upload_host_side_ms": sum(hup) / len(hup) * 1e3,      # (until the build thread has the first chunk: the rest is sent in the background)
                 "first_call_ms": first_host_ms, "gfa_md5": md5_host, "same_graph_as_device_entry": True}
        del text_host

    # ---- T_e2e: the whole command on a FASTA directory (compress.rs:34 -> :49) -----------------------------------------------------
    t_e2e = None
    if world == 1 and mode == "single" and not args.no_e2e and not emu_lib:
        import hashlib
        import shutil
        import subprocess
        import tempfile
        from autocycler_amd import synth
        tmp = Path(tempfile.mkdtemp(prefix="ac_bench_e2e_", dir="/dev/shm" if Path("/dev/shm").is_dir() else None))
        try:
            synth.write_fasta_dir(named[2]() if named else synth.make_assemblies(args.assemblies, genome=args.genome, plasmid=args.plasmid, sub=args.sub,
                                                                                 indel=args.indel, seed=51_000), tmp / "in")
            fasta_bytes = sum(f.stat().st_size for f in (tmp / "in").iterdir())
            threads = min(os.cpu_count() or 8, 32)
            runs = []
            for r in range(3):
                shutil.rmtree(tmp / "out", ignore_errors=True)
                times = (C.c_double * 4)()
                tc = time.perf_counter()
                if lib.ac_compress_dir(str(tmp / "in").encode(), str(tmp / "out").encode(), C.c_uint32(k), C.c_uint32(25), C.c_int(threads),
                                       C.c_int(local_rank), None, times):
                    raise RuntimeError(lib.ac_last_error().decode())
                runs.append({"wall_s": time.perf_counter() - tc, "load_s": times[0], "upload_and_end_repair_s": times[1], "graph_s": times[2], "write_s": times[3]})
            md5_e2e = hashlib.md5((tmp / "out" / "input_assemblies.gfa").read_bytes()).hexdigest()
            best = min(runs, key=lambda x: x["wall_s"])
            t_e2e = {"value": bases / 1e6 / best["wall_s"], "unit": "Mbp/s", **best, "runs_wall_s": [round(x["wall_s"], 4) for x in runs],
                     "what": f"ac_compress_dir in this (warm) process: {args.assemblies} FASTA files ({fasta_bytes / 1e6:.0f} MB, tmpfs) -> "
                             f"input_assemblies.gfa + .yaml, {threads} host threads for load and GFA formatting; best of 3",
                     "gfa_md5": md5_e2e}
            cli = ROOT / "autocycler_amd" / "autocycler-compress"
            if cli.exists():      # the same command as a fresh process: HIP context creation and code-object load included
                shutil.rmtree(tmp / "out", ignore_errors=True)
                tc = time.perf_counter()
                pr = subprocess.run([str(cli), "compress", "-i", str(tmp / "in"), "-a", str(tmp / "out"), "--kmer", str(k), "-t", str(threads),
                                     "--device", str(local_rank)], capture_output=True, text=True)
                wall = time.perf_counter() - tc
                if pr.returncode == 0:
                    t_e2e["cli_fresh_process"] = {"wall_s": wall, "value": bases / 1e6 / wall,
                                                  "gfa_md5": hashlib.md5((tmp / "out" / "input_assemblies.gfa").read_bytes()).hexdigest(),
                                                  "stage_line": next((l for l in pr.stderr.splitlines() if l.startswith("Stage times")), "")}
                else:
                    t_e2e["cli_fresh_process"] = {"error": pr.stderr[-300:]}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        b = torch.tensor([bases], dtype=torch.float64, device=dev)
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        total_bases = float(b.item())
    else:
        total_bases = float(bases)

    graph_info = {**g.stats_post, "kmers": g.kmer_count}
    # Secondary figure at N > 1 (not `value`): the same ranks running their slices as N independent compress jobs — no
    # data-path collective — timed the same way, so that the cost of making it ONE job is visible in the same line.
    independent = None
    if world > 1 and mode == "sharded" and not args.no_independent:
        def step_ind():
            h = C.c_void_p()
            rc = lib.ac_compress_build_device(C.c_uint32(k), C.c_uint32(args.assemblies), C.c_void_p(d_text.data_ptr()),
                                              C.c_uint64(n_text), off, lens, ids, d1, d2, C.c_uint32(n), C.c_int(local_rank), C.byref(h))
            if rc:
                raise RuntimeError(lib.ac_last_error().decode())
            return _capi.Graph(lib, h, n)
        g.close(); g = None
        for _ in range(max(args.warmup, 1)):
            step_ind().close()
        barrier()
        t_i = time.perf_counter()
        for _ in range(args.steps):
            step_ind().close()
        barrier()
        e_i = torch.tensor([time.perf_counter() - t_i], dtype=torch.float64, device=dev)
        dist.all_reduce(e_i, op=dist.ReduceOp.MAX)
        independent = {"value": total_bases / 1e6 / (float(e_i.item()) / args.steps), "unit": "Mbp/s",
                       "ms_per_step": float(e_i.item()) / args.steps * 1e3,
                       "what": "the same ranks and texts as N independent compress jobs (one per GPU, no data-path collective)"}

    # Secondary figure at N > 1 (not `value`): the same job through the drop-in boundary a Rust caller would use — ONE process driving all N
    # devices through ac_compress_build_multi (in-library RCCL).  It runs in a subprocess of rank 0 while the other ranks wait (they give their
    # device memory back first), so whatever happens to it cannot take the benchmark line with it.
    in_library = None
    if world > 1 and mode == "sharded" and not args.no_multi_entry and not emu_lib:
        if g is not None:
            g.close(); g = None
        del d_text
        shard = None
        torch.cuda.empty_cache()
        lib.ac_release_memory()
        barrier()
        if rank == 0:
            import subprocess
            try:
                pr = subprocess.run([sys.executable, str(ROOT / "tools" / "multi_bench.py"), "--bench-job", str(world), "--assemblies", str(args.assemblies),
                                     "--genome", str(args.genome), "--steps", str(min(args.steps, 5)), "--warmup", "2"],
                                    env={**os.environ, "AC_NO_TORCH": "1"}, capture_output=True, text=True, timeout=600)
                rows = [l for l in pr.stdout.splitlines() if l.startswith("{")]
                in_library = json.loads(rows[-1]) if rows else {"error": (pr.stderr or "no output")[-400:]}
            except Exception as e:      # noqa: BLE001 — a timeout or a crash of the subprocess is reported, not raised
                in_library = {"error": repr(e)[:400]}
        barrier()

    if rank == 0:
        # The headline: SURVEY.md 8(d)'s T_hot bracket (host RAM -> host RAM through ac_compress_build) wherever it was timed (N = 1);
        # the device-resident bracket is the extra key `hbm_resident`.  N > 1 (and --no-host-bracket): the device-resident steps.
        hbm_resident = {"value": total_bases / 1e6 / (elapsed / args.steps), "unit": "Mbp/s", "ms_per_step": elapsed / args.steps * 1e3, "steps": args.steps,
                        "timed_region": "padded+repaired text RESIDENT IN HBM -> final unitig graph in host RAM (ac_compress_build_device: device build + D2H)",
                        "step_ms_list": [round(x * 1e3, 2) for x in step_s],
                        "step_ms": {"min": min(step_s) * 1e3, "median": sorted(step_s)[len(step_s) // 2] * 1e3, "max": max(step_s) * 1e3},
                        "total_device_timed_ms": sum(t["total_device"] for t in tms) / len(tms) * 1e3}
        headline_is_hot = t_hot is not None
        if headline_is_hot:
            elapsed_head, tms_head, step_head = t_hot["elapsed_s"], tms_hot, t_hot["step_ms_list"]
        else:
            elapsed_head, tms_head, step_head = elapsed, tms, hbm_resident["step_ms_list"]
        ms_per_step = elapsed_head / args.steps * 1e3
        value = total_bases / 1e6 / (elapsed_head / args.steps)
        ins_ms = sum(t["insert_kernel_ms"] for t in tms_head) / len(tms_head)
        if emu_lib and ins_ms <= 0:
            ins_ms = 1e-3      # the emulation has no HIP events
        alg_bytes = A_K(k) * bases      # one (key, tag) record written and read per input base — kept as `whole_path_equiv` only
        # HBM bytes per build of the big kernels from the PMC passes (collected separately with tools/pmc_lean.sh and committed under
        # profiles/; rocprofv3 counters cannot be read from inside this process).  Only quoted when this run is the workload the
        # counters were collected on.
        pj, traffic_src, pmc_note = None, None, None
        pmc = ROOT / "profiles" / (f"pmc_traffic_{args.workload}.json" if named else "pmc_traffic.json")
        default_workload = (args.assemblies, args.genome, args.plasmid, args.sub, args.indel, k) == (96, 5_000_000, 100_000, 1e-4, 1e-5, 51) and not named
        if emu_lib and os.environ.get("BENCH_EMU_ASSUME_DEFAULT"):
            default_workload = True      # dry run only: exercise the fields that are quoted for the default workload
        counted_workload = default_workload or bool(named)      # the workloads tools/pmc_lean.sh can rebuild: config C and the named ones
        lib.ac_source_hash.restype = C.c_char_p
        lib_hash = lib.ac_source_hash().decode()
        if counted_workload and args.pmc != "off":
            if pmc.exists() and args.pmc != "live":
                cand = json.loads(pmc.read_text())
                if cand.get("source_hash") == lib_hash or (emu_lib and os.environ.get("BENCH_EMU_ASSUME_DEFAULT")):
                    pj = cand
                    traffic_src = (f"profiles/{pmc.name}: rocprofv3 PMC passes (tools/pmc_lean.sh) on a library built from exactly these sources "
                                   f"(source hash {lib_hash}); FETCH_SIZE x2 per the gfx950 calibration, + WRITE_SIZE")
                else:
                    pmc_note = (f"profiles/{pmc.name} is stale: collected on sources {cand.get('source_hash')}, this library is {lib_hash}")
            if pj is None and args.pmc in ("auto", "live") and not emu_lib and world == 1:
                live = collect_pmc_live(n_text, workload=args.workload)
                if "error" in live:
                    pmc_note = ((pmc_note + "; ") if pmc_note else "") + "live collection failed: " + live["error"]
                else:
                    pj = live
                    traffic_src = ("counted in this run: two rocprofv3 PMC passes (tools/pmc_lean.sh: FETCH_SIZE, WRITE_SIZE, --kernel-trace only) of the torch-free "
                                   "driver building this workload with this library; FETCH_SIZE x2 per the gfx950 calibration, + WRITE_SIZE")
        stage = {key: sum(t[key] for t in stage_tms) / len(stage_tms) for key in
                 ("pack", "insert", "collect_sort", "degree", "segment", "minkey", "rank", "links", "paths", "seqs", "analysis", "expand", "finalize", "d2h",
                  "total_device") + (("fragments", "union_pack", "union_insert") if mode == "sharded" else ())}
        sharding = {"single": "one device", "independent": "by assembly set, one unrelated compress job per GPU",
                    "sharded": "ONE job: sequences sharded by rank, the k-mer table partitioned over the ranks by key hash (owner = hash of the "
                               "canonical middle): all-gather of novel-run fragments, SUM all-reduces of the owners' bitmap / degree / link "
                               "contributions, walk-start keys all-gathered and answered by their owners, per-unitig all-reduce; unitigs + "
                               "links end in rank 0's host RAM, the paths (P lines) " +
                               ("of all sequences too (gathered)" if args.gather_paths else "of each rank's sequences in that rank's host RAM") +
                               " (autocycler_amd/sharded.py)"}[mode]
        line = {
            "metric": "Mbp/sec through compress->unitig GFA (k=%d)" % k,
            "value": value, "unit": "Mbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "value_bracket": ("t_hot: host RAM -> host RAM through ac_compress_build" if headline_is_hot else
                              "hbm_resident: text resident in HBM -> graph in host RAM (N > 1: the torch driver keeps the ranks' texts on their devices; --no-host-bracket)"),
            "ms_per_step": ms_per_step, "ms_per_step_median": sorted(step_head)[len(step_head) // 2], "ms_per_step_max": max(step_head),
            "value_from_median_step": total_bases / 1e3 / sorted(step_head)[len(step_head) // 2],
            "value_note": "value / ms_per_step = the whole timed bracket (K steps between two barriers) / K, as the benchmark contract asks; the median and the slowest step beside it",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic" if not emu_lib else "emulation dry run (not a measurement)",
            "config": {"workload": (f"{args.workload}: {_synth.WORKLOAD_NOTES.get(args.workload, 'a named workload')} (autocycler_amd/synth.py WORKLOADS; sub {args.sub:g}, indel {args.indel:g})"
                                    if named else
                                    f"{args.assemblies} x ~{args.genome / 1e6:g} Mbp synthetic assemblies per GPU (+plasmid {args.plasmid} bp, "
                                    f"sub {args.sub:g}, indel {args.indel:g}), k={k}, " +
                                   (f"1 species; {workload_label(args, k)}" if world == 1 else
                                    (f"ONE job of {world} species, one per GPU (mixed-species job as in BASELINE.json configs[4]; rank 0 holds "
                                     "exactly the N=1 workload, configs[2])" if (mode == "sharded" and args.species == "per-gpu") else
                                     f"ONE job of {world * args.assemblies} assemblies of one species" if mode == "sharded" else
                                     f"{world} unrelated jobs of one species each"))),
                       "bases_per_gpu": bases, "sequences_per_gpu": n, "mode": mode, "sharding": sharding,
                       "timed_region": ("SURVEY.md 8(d) T_hot: padded+repaired sequences in pageable host RAM -> final unitig graph in host RAM through "
                                        "ac_compress_build (host-side 2-bit pack, H2D, device build, D2H); the same region with the text already resident "
                                        "in HBM is `hbm_resident`, the whole command `t_e2e`") if headline_is_hot else
                                       ("padded+repaired sequences in HBM -> final unitig graph in host RAM (device build + D2H)")},
            "roofline": None,      # (filled in below)
            "hbm_resident": hbm_resident,
            "t_hot": t_hot, "t_e2e": t_e2e, "cold_first_build_ms": cold_first_build_ms, "init_builds_run": init_builds_run,
            "untimed_builds_before_the_timed_steps": {"init": init_builds_run, "warmup_device_entry": args.warmup, "device_entry_steps": args.steps + 2,
                                                      "first_host_entry_call": 1 if headline_is_hot else 0, "warmup_host_entry": max(args.warmup, 1) if headline_is_hot else 0},
            "total_device_timed_ms": sum(t["total_device"] for t in tms_head) / len(tms_head) * 1e3,
            "step_ms_list": step_head,
            "step_ms": {"min": min(step_head), "median": sorted(step_head)[len(step_head) // 2], "max": max(step_head)},
            "launches_per_build": tms[-1].get("launches"), "host_round_trips_per_build": tms[-1].get("readbacks"),
            "stages_s": stage, "stages_note": "from 2 extra untimed builds with per-stage stream syncs (total_device there includes them)",
            "graph": {**graph_info, "distinct_canonical": tms[-1]["n_distinct"],
                      "path_entries": tms[-1]["n_path_entries"], "table_capacity": tms[-1]["table_capacity"],
                      "simplify_passes": tms[-1]["simplify_passes"]},
            "prep_s": {"generate": t_gen, "end_repair": t_repair, "end_repair_info": repair_info, "h2d": t_h2d, "h2d_GBps": n_text / t_h2d / 1e9},
        }
        # ---- roofline ---------------------------------------------------------------------------------------------------------------------
        # Per kernel: `traffic` = HBM-side bytes per build from the PMC passes, `frac` = traffic / time / 8 TB/s (a measured fraction, <= 1);
        # `needed_bytes` = a LOWER bound on the bytes this algorithm has to move (model stated in `needed_model`), `achieved` = needed_bytes /
        # time, `waste` = traffic / needed_bytes (re-reads and partial lines); `ceiling` = what actually binds the kernel, by name, with the
        # share of the kernel's time the unavoidable operations of that kind take at the device's measured rate.  The SURVEY.md 8(d) figure
        # (49 B/bp at k = 51: a (key, tag) record written and read per input base) is only kept as `whole_path_equiv`: the run-following
        # insert never materialises those records, so dividing them by the insert's time gives a "fraction" above 1 (round 2: 3.13).
        cas, rd = C.c_double(), C.c_double()
        st = tms_head[-1]
        # (measured on a table of the size this workload's k-mer table has: beyond the 256 MB Infinity Cache the claim rate drops)
        ceil_slots = int(st.get("table_capacity") or (1 << 24))
        have_ceil = lib.ac_random_access_ceilings_at(C.c_int(local_rank), C.c_uint64(ceil_slots), C.byref(cas), C.byref(rd)) == 0 and cas.value > 0
        claims = st["n_local_distinct"] or st["n_distinct"]
        U_now = graph_info["unitigs"]
        per_kernel = (pj or {}).get("per_kernel_per_build", {})
        pmc_of = lambda pat: sum(v["hbm_side_bytes"] for kk, v in per_kernel.items() if pat in kk) or None

        def roof(kernel, what, ms, needed, model, traffic, ceiling):
            r = {"bound": "hbm", "kernel": kernel, "what": what, "kernel_ms": ms, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                 "needed_bytes": needed, "needed_model": model, "achieved": needed / (ms * 1e-3) / 1e9 if ms else None,
                 "frac_needed": needed / (ms * 1e-3) / HBM_PEAK if ms else None,
                 "traffic": traffic, "traffic_source": traffic_src if traffic else None,
                 "traffic_GBps": traffic / (ms * 1e-3) / 1e9 if (traffic and ms) else None,
                 "waste": traffic / needed if traffic else None, "ceiling": ceiling}
            # the fraction of the HBM peak: measured bytes where the counters exist for this workload, else the model's lower bound
            r["frac"] = (traffic / (ms * 1e-3) / HBM_PEAK) if (traffic and ms) else r["frac_needed"]
            r["frac_source"] = "traffic (PMC) / kernel time / peak" if traffic else "needed_bytes (model, a lower bound) / kernel time / peak"
            return r

        packed_b = 0.375 * n_text      # 2 bits of code + 1 mask bit per text position
        ins_needed = 2 * packed_b + 64.0 * claims + n_text / 8.0
        ins_ceiling = None
        if have_ceil:
            ins_ceiling = {"name": "cas", "rate_Gops": cas.value, "operations": claims, "bound_ms": claims / cas.value / 1e6,
                           "frac": claims / cas.value / 1e6 / ins_ms,
                           "table_slots": ceil_slots,
                           "note": "one atomicCAS per distinct k-mer at the device's measured random-CAS rate on a table of this workload's table size: "
                                   "the share of the kernel's time that no insert into a hash table with random placement can avoid"}
        line["roofline"] = roof(
            "insert_wave_kernel<W> (wavefront-cooperative run-following k-mer insert; %d phase launches per build)" % st["insert_launches"],
            "KmerGraph::add_sequences (kmer_graph.rs:86-134) + iterate_kmers' key set", ins_ms, ins_needed,
            "packed text (0.375 B per position) read on both sides of a followed run + one 64-byte slot line per distinct k-mer + the novel bitmap "
            "(1 bit per position)", pmc_of("insert_wave_kernel"), ins_ceiling)
        line["roofline"]["library_source_hash"] = lib_hash
        if pj and pj.get("traffic_raw"):      # the raw counter, the streaming-calibrated upper bound and the factor actually applied
            raw, x2 = pj["traffic_raw"], pj.get("traffic_streaming_x2") or pj["traffic_raw"]
            line["roofline"]["traffic_raw"] = raw
            line["roofline"]["traffic_streaming_x2"] = x2
            line["roofline"]["fetch_factor_applied"] = pj.get("fetch_correction")
            line["roofline"]["fetch_factor_calibration"] = pj.get("calibration_random") or "streaming (PackFunctor) only: no random-access calibration file for these sources"
            line["roofline"]["frac_range"] = [raw / (ins_ms * 1e-3) / HBM_PEAK, x2 / (ins_ms * 1e-3) / HBM_PEAK]
        if pmc_note:
            line["roofline"]["traffic_note"] = pmc_note
        line["roofline"]["whole_path_equiv"] = {"bytes": alg_bytes, "B_per_bp": A_K(k), "frac_of_peak_over_the_whole_step": alg_bytes / (elapsed_head / args.steps) / HBM_PEAK,
                                                "note": "SURVEY.md 8(d) accounting (a sort-based design's record traffic) over the WHOLE timed step, not over the insert kernel"}
        others = []
        n_ent = st["n_path_entries"]
        walk_ms = stage["paths"] * 1e3
        walk_ceiling = None
        if have_ceil:
            n_walkers = n_text / 256.0
            ops = n_ent + 9.0 * n_walkers      # one successor-table gather per entry + the ~9 dependent lines of each walker's start-up lookup
            walk_ceiling = {"name": "random_read", "rate_Gops": rd.value, "operations": ops, "bound_ms": ops / rd.value / 1e6,
                            "frac": ops / rd.value / 1e6 / walk_ms if walk_ms else None,
                            "note": "one dependent gather per path entry plus each walker's start-up lookup, at the device's measured random 8-byte read rate"}
        if st.get("path_runs_copied", 0) > 0:
            # the copying walk (chosen by the library's cost model on large redundant texts): the text between the insert's
            # followed runs is walked, the runs' entries are copied from the walked entries of the stretch they repeat
            copy_kernels = ("PathWalkFunctor", "RunOutFunctor", "GapOutFunctor", "WalkCompactFunctor", "RunRangeFunctor", "RunFilterFunctor",
                            "RunGatherFunctor", "RunCompactFunctor", "GapWalkersFunctor", "WalkerRangeFunctor", "SegCountFunctor", "WlinkFlagFunctor",
                            "WalkInfoFunctor", "MaybeDestFunctor", "PathOffCopyFunctor", "PathEndsFunctor")
            walked = st["path_entries_walked"]
            n_steps = 28      # ~24 dependent launches + 4 read-backs in the stage (profiles/r09e_timeline_path_copy_rows.txt)
            others.append(roof("copying path walk: PathWalkFunctor<W> over the gaps + RunOutFunctor / GapOutFunctor + run bookkeeping (stage `paths`)",
                               "get_unitig_path_for_sequence (unitig_graph.rs:407-465) for all sequences",
                               walk_ms, 8.0 * n_ent + 24.0 * st["path_runs_copied"] + packed_b * (walked / max(n_ent, 1)) + 96.0 * U_now,
                               "every path entry read once from the walked entries and written once (2 x 4 B) + 24 B per copied run + the walked share of "
                               "the packed text read once + the unitig / successor records (96 B per unitig) read once",
                               sum(pmc_of(nm) or 0 for nm in copy_kernels) or None,
                               {"name": "dependent_launches", "operations": n_steps, "bound_ms": n_steps * 0.010, "frac": n_steps * 0.010 / walk_ms if walk_ms else None,
                                "note": "kernels and host read-backs that must follow one another (scan -> size -> launch), ~10 us each: the stage's floor once "
                                        "its per-entry work is streaming copies; %d of %d entries were walked, the rest copied" % (walked, n_ent)}))
        else:
            others.append(roof("PathWalkFunctor<W> + PathCompactFunctor (stage `paths`)", "get_unitig_path_for_sequence (unitig_graph.rs:407-465) for all sequences",
                               walk_ms, 8.0 * n_ent + packed_b + 96.0 * U_now,
                               "every path entry written to its staging slot and once more in text order (2 x 4 B) + the packed text read once + the unitig / "
                               "successor records (96 B per unitig) read once",
                               (pmc_of("PathWalkFunctor") or 0) + (pmc_of("PathCompactFunctor") or 0) or None, walk_ceiling))
        exp_ms = stage["expand"] * 1e3
        n_cand = st["n_candidates"]
        n_launch = st["n_levels"] * st["simplify_passes"]
        others.append(roof("expand_wave_kernel<W,16> + level scheduling + sequence rewrite (stage `expand`)",
                           "expand_repeats (graph_simplification.rs:43-86)", exp_ms, n_cand * (120.0 + 5 * 64.0) + 2.0 * graph_info["total_length"],
                           "per candidate junction once: its link row and five source records (120 B) + one 64-byte line of each source's sequence; the "
                           "sequences rewritten once after the last pass (read + write)",
                           pmc_of("expand_wave_kernel"),
                           {"name": "dependent_launches", "operations": n_launch, "bound_ms": n_launch * 0.010, "frac": n_launch * 0.010 / exp_ms if exp_ms else None,
                            "note": "levels x passes kernels that must run one after the other (the reference's visiting order), ~10 us each: junction chains, "
                                    "not bytes, bind this stage"}))
        line["roofline_other"] = others
        # The kernels of this path are bound by random accesses into the k-mer table, not by streamed bytes: price them against
        # the device's measured random-access ceilings as well (a ~20 ms microbenchmark inside the library).
        if have_ceil:
            deg_ms = stage["degree"] * 1e3
            line["random_access"] = {
                "cas_ceiling_Gops": cas.value, "read_ceiling_Gops": rd.value, "measured_on_table_slots": ceil_slots,
                "insert": {"slot_claims": st["n_local_distinct"] or st["n_distinct"], "kernel_ms": ins_ms,
                           "claims_only_bound_ms": (st["n_local_distinct"] or st["n_distinct"]) / cas.value / 1e6,
                           "frac_of_cas_ceiling": (st["n_local_distinct"] or st["n_distinct"]) / cas.value / 1e6 / ins_ms},
                "degree": {"stage_ms": deg_ms, "walks_if_every_degree_were_probed": 2 * st["n_distinct"],
                           "walks_only_bound_ms": 2 * st["n_distinct"] / rd.value / 1e6,
                           "stage_over_that_bound": deg_ms / (2 * st["n_distinct"] / rd.value / 1e6)},
                "note": "bounds count only the unavoidable random accesses: one CAS per distinct k-mer for the insert; for the degree pass "
                        "the two probe-cluster walks per distinct k-mer it WOULD need if every degree were probed (round 1-2 form) — the "
                        "single-device pass settles nearly all of them from the sibling bits the insert collects (DESIGN.md section 4, K5) and only "
                        "probes the rest, which is why the stage can be faster than that bound"}
        if independent is not None:
            line["independent_jobs"] = independent
        if in_library is not None:
            line["in_library_multi"] = in_library
        if mode == "sharded":
            line["sharded"] = {**last_info, "fragments_rank0": tms[-1]["n_fragments"], "fragment_bytes_rank0": tms[-1]["fragment_bytes"],
                               "local_distinct_rank0": tms[-1]["n_local_distinct"]}
        if os.environ.get("BENCH_STAGE_TIMING"):
            line["timed_stage_ms"] = [{kk: round(vv * 1e3, 2) for kk, vv in t.items() if isinstance(vv, float) and vv > 2e-4 and kk != "insert_kernel_ms"} for t in tms]
        if not args.no_cpu_baseline and world == 1:
            a, b2 = args.cpu_sample.split("x")
            sample = cpu_baseline(k, int(a), int(b2))      # timed now, on this box's host cores
            # cpu_baseline.value = what was TIMED IN THIS RUN, on this box's host cores (the figure recorded elsewhere is context, under
            # its own key; a speed-up quoted from `value` compares two clocks of one machine)
            sample["host_threads_in_t_hot"] = int(os.environ.get("AC_UPLOAD_THREADS", "24"))      # the GPU path's T_hot packs the text with this many host threads; the CPU path's hot stages are single-threaded like the reference's
            line["cpu_baseline"] = sample
            gold = ROOT / "tests" / "golden" / (f"{args.workload}.json" if named else "configC_k51.json")      # the oracle on the WHOLE workload, run once where it was recorded
            if counted_workload and gold.exists():
                try:
                    gj = json.loads(gold.read_text())
                    hot = gj["seconds"]["kmer_graph"] + gj["seconds"]["unitig_graph"] + gj["seconds"]["simplify"]
                    md5_now = (t_hot or {}).get("gfa_md5")
                    if md5_now is not None and not emu_lib and md5_now != gj["gfa_md5"]:
                        raise SystemExit(f"this run's GFA md5 {md5_now} is not the oracle's full-size golden {gj['gfa_md5']} ({gold.name})")
                    # (the small sample has 4 copies of every k-mer instead of 96 and understates the CPU path ~1.8x: the whole-workload figure,
                    # recorded once on another host, stays beside it)
                    line["cpu_baseline"]["recorded_whole_workload"] = {
                        "value": (487_499_962 if emu_lib else bases) / 1e6 / hot, "unit": "Mbp/s", "cores": 1, "kind": "port", "seconds": hot,
                        "what": f"the WHOLE workload ({args.workload or 'configC_k51: 96 x ~5 Mbp, k=51'}) through the C++ restatement of the reference CPU path, hot stages on 1 core: "
                                "recorded once by tests/golden/make_configC_golden.sh on " + gj.get("host", "the build container") + " (NOT timed in this run)",
                        "gfa_md5": gj["gfa_md5"],
                        "gfa_md5_equals_this_runs": (md5_now == gj["gfa_md5"]) if md5_now is not None else None}
                except (KeyError, ValueError, TypeError):
                    pass
        print(json.dumps(line))
    if g is not None:
        g.close()
    lib.ac_seqs_free(h_seqs)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
