#!/usr/bin/env python3
"""bench.py -- throughput of the coupling-layer hot path on MI355X.

Workload (BASELINE.json configs[1], the one `metric` is quoted on): 32 x [CoupledRationalQuadraticSpline(64, 2
blocks, 128 hidden, K=8, tails linear, tail_bound 3) + LULinearPermute(64)], DiagGaussian(64) base, fp32, batch
65 536 synthetic N(0, I) rows per GPU.  A "step" is one `log_prob` pass of the whole flow over the batch (every
layer's inverse + log|det J| accumulation + base log-density), inputs resident in HBM.  Weights: seeded
construction (bit-identical to constructing the reference with the same seed, tests/test_state_dict_compat.py)
plus the sigma = 0.01 parameter perturbation of SURVEY.md section 8d so that no layer is the identity.

    python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one rank per GPU with torch.distributed.run; the batch is sharded by rows
(weak scaling: 65 536 rows per rank), the only collective is the 8-byte NLL all-reduce, inside every timed step.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DIM, HIDDEN, BLOCKS, BINS, LAYERS, BATCH = 64, 128, 2, 8, 32, 65536
SIGMA = 0.01


def build_c2_model(num_layers=LAYERS, dim=DIM, hidden=HIDDEN, blocks=BLOCKS, bins=BINS, seed=0, sigma=SIGMA, lib=None):
    """The benchmark model.  `lib` may be the reference package (tests only) -- same constructor calls."""
    if lib is None:
        import normflows_amd as lib
    torch.manual_seed(seed)
    flows = []
    for _ in range(num_layers):
        flows += [lib.flows.CoupledRationalQuadraticSpline(dim, blocks, hidden, num_bins=bins)]
        flows += [lib.flows.LULinearPermute(dim)]
    q0 = lib.distributions.DiagGaussian(dim, trainable=False)
    model = lib.NormalizingFlow(q0, flows)
    perturb_(model, sigma)
    return model


def perturb_(model, sigma, seed=1234):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(sigma * torch.randn(p.shape, generator=g, dtype=p.dtype))
    return g


def c2_inputs(batch=BATCH, dim=DIM, seed=1234, rank=0):
    g = torch.Generator().manual_seed(seed + 7919 * (rank + 1))
    return torch.randn(batch, dim, generator=g)


def state_to_numpy(model):
    return {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


# ---- algorithmic work per sample of one full pass ------------------
def c2_flops_per_sample(dim=DIM, hidden=HIDDEN, blocks=BLOCKS, bins=BINS, layers=LAYERS):
    nI = (dim + 1) // 2
    nT = dim // 2
    out = nT * (3 * bins - 1)
    cond = 2 * (nI * hidden + 2 * blocks * hidden * hidden + hidden * out)
    lu = 2 * 2 * dim * dim  # two dense 64x64 mat-vecs, SURVEY.md 8d
    return layers * (cond + lu)


def c2_bytes_per_sample(dim=DIM, bins=BINS, layers=LAYERS, fused=False):
    nT = dim // 2
    rqs = 2 * dim * 4 + 8 + (0 if fused else nT * (3 * bins - 1) * 4)
    lu = 2 * dim * 4 + 8
    return layers * (rqs + lu) + dim * 4 + 4


def _events_ms(fn, reps):
    """Average duration (ms) of fn() measured with HIP events on torch's current stream (the launch stream)."""
    start = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    stop = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    for i in range(reps):
        start[i].record()
        fn()
        stop[i].record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in zip(start, stop))
    return sum(ts) / len(ts), ts[len(ts) // 2]


def kernel_breakdown(model, x, reps=5):
    """HIP-event timing of the launches of one eager log_prob pass (instrumented run, NOT the timed region).
    All fused [LULinearPermute + CoupledRQS] layer pairs of the model run as ONE persistent launch of
    nf::rqs_fused_kernel<0, true, false, 8> (core.run_chain -> nf_rqs_fused_chain); it is bracketed by one event pair recorded on
    torch's current stream (= the stream the C ABI launches on).  Returns {name: (avg_ms, launches_per_pass, layers)}."""
    import normflows_amd as nfa
    from normflows_amd.core import run_chain
    flows = list(model.flows)
    npairs = sum(1 for i in range(1, len(flows)) if isinstance(flows[i], nfa.flows.LULinearPermute)
                 and isinstance(flows[i - 1], nfa.flows.CoupledRationalQuadraticSpline)
                 and flows[i - 1]._pair_eligible(x, flows[i]))
    out = {}

    def chain():
        log_q = torch.zeros(len(x), device=x.device)
        return run_chain(flows, x, True, log_q, +1), log_q

    z, log_q = chain()  # one-off weight packing stays outside the timed launches
    avg, med = _events_ms(lambda: chain(), reps)
    out["rqs_fused_chain"] = (med, 1 if npairs == len(flows) // 2 else None, npairs)
    avg, med = _events_ms(lambda: model.q0._log_prob_acc(z, log_q, +1), reps)
    out["diag_gaussian"] = (med, 1, 0)
    return out


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/*_pmc.json, produced by tools/summarize_profiles.py: FETCH_SIZE and WRITE_SIZE collected in separate
    passes, KB -> bytes, FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM).  None when no summary is present;
    counters cannot be read from inside the timed process."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), key=lambda q: ("chain" in q, q))
    for path in reversed(cands):  # the persistent-chain profile (the kernel as benchmarked) first
        try:
            d = json.load(open(path))
            if "rqs_fused" in d.get("kernel", "") and "hbm_traffic_bytes" in d.get("derived", {}):
                return d["derived"]["hbm_traffic_bytes"], "profiles/%s (static: committed rocprofv3 --pmc passes, not this run)" % os.path.basename(path)
        except Exception:
            pass
    return None, None


def reference_cpu_leg():
    """The `kind: "reference"` leg: normflows itself (PyTorch CPU) timed by tools/cpu_reference.py.  /root/reference does not exist
    on the GPU box at bench time, so the newest committed JSON is reported with where it was measured: rounds 5-6 staged the package
    for one gpurun call and timed it ON THE GPU BOX's host cores (profiles/r06_cpu_reference_gpubox.json: thread sweep 8 ... 256, each
    setting in its own process under a time limit -- 128 and 256 threads do not finish and are recorded as such); rounds 2-4: the
    8-core build container (profiles/r02_cpu_reference.json)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*cpu_reference*.json")), reverse=True):
        try:
            d = json.load(open(path))
            leg = {"value": d["log_prob"]["samples_per_s"], "unit": "samples/s", "cores": d["cores"], "kind": "reference",
                   "where": d.get("where", "build container"), "host_threads": d.get("host_threads", d["cores"]),
                   "cpu_model": d.get("cpu_model"), "torch": d.get("torch"),
                   "sample": "normflows (PyTorch CPU, %d threads%s) log_prob of the same model on %d benchmark rows, best of %d "
                             "(%.1f s); tools/cpu_reference.py" % (d["torch_threads"],
                                                                   ", best of the sweep %s" % [e["torch_threads"] for e in d["sweep"]] if d.get("sweep") else "",
                                                                   d["rows"], d["repeats"], d["log_prob"]["best_s"]),
                   "nll_nats_per_dim": d["log_prob"]["nll_nats_per_dim"], "source": "profiles/" + os.path.basename(path)}
            if "sample" in d:
                leg["sample_direction_samples_per_s"] = d["sample"]["samples_per_s"]
            if d.get("sweep"):
                leg["thread_sweep_samples_per_s"] = {str(e["torch_threads"]): (round(e["samples_per_s"], 1) if e.get("samples_per_s")
                                                                               else "did not finish in %.0f s" % e.get("timeout_s", 0))
                                                     for e in d["sweep"]}
            return leg
        except Exception:
            pass
    return None


def cpu_baseline(model, rows, gpu_lp=None, gpu_lp3=None):
    """The CPU oracle (a port of the reference algorithm) timed on this host on a bounded sample of the same workload:
    log_prob of the same 32-layer model on `rows` benchmark rows through ONE C call (oracle/nf_oracle.c
    nfo_nsf_log_prob: OpenMP over 64-row chunks, every chunk runs the whole 64-layer chain)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nf_oracle
    ora = nf_oracle.OracleNSF(state_to_numpy(model), num_layers=len(model.flows), K=BINS, tail_bound=3.0)
    x = c2_inputs(rows, DIM).numpy()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = int(os.environ.get("OMP_NUM_THREADS", cores))
    ora.log_prob_whole(x[:256])  # warm-up (thread pool, page-in)
    t0 = time.perf_counter()
    lp = ora.log_prob_whole(x)
    dt = time.perf_counter() - t0
    res = {"value": rows / dt, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": "log_prob of the same %d-layer model on %d rows (oracle/nf_oracle.c nfo_nsf_log_prob, OpenMP, %.1f s)"
                     % (len(model.flows) // 2, rows, dt), "nll_nats_per_dim": float(-lp.mean() / DIM)}
    # accuracy of the GPU results against the oracle in DOUBLE precision on the first rows of the benchmark batch
    # (the fp32 oracle itself is shown for scale): relative error of log_prob, max over the rows
    n64 = min(2048, rows)
    ref64 = ora.log_prob(x[:n64].astype(np.float64))
    rel = lambda a: float(np.max(np.abs(a.astype(np.float64) - ref64) / np.maximum(1.0, np.abs(ref64))))
    acc = {"rows": n64, "oracle_f32": rel(lp[:n64])}
    if gpu_lp is not None:
        acc["gpu_exact_f32"] = rel(gpu_lp[:n64])
    if gpu_lp3 is not None:
        acc["gpu_bf16x3"] = rel(gpu_lp3[:n64])
    res["max_rel_err_log_prob_vs_fp64_oracle"] = acc
    return res


class CollectiveCounter:
    """Counts the data-path collectives issued inside a `with` block (calls, payload bytes): EVERY torch.distributed collective
    entry point is wrapped, not only all_reduce -- a contract test on `collectives_per_step` must not pass because the code under
    test moved to all_gather -- and the originals are restored on exit, exception or not."""
    NAMES = ("all_reduce", "all_gather", "all_gather_into_tensor", "reduce", "broadcast", "reduce_scatter", "reduce_scatter_tensor",
             "all_to_all", "all_to_all_single", "gather", "scatter", "all_reduce_coalesced", "all_gather_coalesced")

    def __init__(self, active=True):
        self.active, self.coll, self.saved = active, {"calls": 0, "bytes": 0, "by_op": {}}, {}

    def _wrap(self, name, fn):
        def counted(*a, **k):
            nbytes = 0
            for t_ in list(a) + list(k.values()):
                for u in (t_ if isinstance(t_, (list, tuple)) else (t_,)):
                    if torch.is_tensor(u):
                        nbytes = max(nbytes, u.numel() * u.element_size())
            self.coll["calls"] += 1
            self.coll["bytes"] += nbytes
            self.coll["by_op"][name] = self.coll["by_op"].get(name, 0) + 1
            return fn(*a, **k)
        return counted

    def __enter__(self):
        import torch.distributed as dist
        if self.active:
            for name in self.NAMES:
                fn = getattr(dist, name, None)
                if fn is not None:
                    self.saved[name] = fn
                    setattr(dist, name, self._wrap(name, fn))
        return self

    def __exit__(self, *exc):
        import torch.distributed as dist
        for name, fn in self.saved.items():
            setattr(dist, name, fn)
        self.saved = {}
        return False


def make_train_step(model, x, world=1, group=None):
    """One training step of the benchmark model as a closure: forward_kld (core.py:87-102) + backward + [gradient average over the
    ranks, overlapped with backward] + Adam.  Round 6: the parameters are views of ONE flat tensor (dp.FlatParameters): the backward
    kernels write every gradient into one flat buffer, torch.optim.Adam(fused=True) steps one tensor (one launch instead of 17) and
    the data-parallel all-reduces run on slices of that buffer in place."""
    from normflows_amd import dp
    model.use_graphs(False)
    flat = dp.FlatParameters(model)
    opt = torch.optim.Adam(flat.parameters(), lr=1e-4, fused=True)
    avg = dp.OverlappedGradientAverager(flat.params, group=group, bucket_bytes=8 << 20, flat=flat) if world > 1 else None

    def one_step():
        flat.zero_grad()
        loss_ = model.forward_kld(x)          # -mean(log_q) of the rank's rows; equal shards: the average of the ranks' gradients
        loss_.backward()                      # is the gradient of the global mean
        if avg is not None:
            avg.finish()
        flat.sync()
        opt.step()
        return loss_
    return one_step, flat, opt


def train_step_record(model, x, world=1, steps=10):
    """The `train_step` record of `secondary` (N = 1): median of individually synchronised steps and the mean of back-to-back ones."""
    one_step, flat, opt = make_train_step(model, x, world)
    ts = []
    for i in range(3 + steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = one_step()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]     # median of 10 individually synchronised steps (the mean is sensitive to allocator warm-up)
    torch.cuda.synchronize()   # ... and the same 10 steps back to back, as a training loop runs them (no sync in between)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flop = 3 * c2_flops_per_sample(layers=len(model.flows) // 2) * x.shape[0]
    rec = {"workload": "forward_kld + backward + Adam on the benchmark model and batch", "ms_per_step": dt * 1e3,
           "optimizer": "torch.optim.Adam(lr=1e-4, fused=True) on dp.FlatParameters(model).parameters() (one flat tensor)",
           "statistic": "mean of 10 back-to-back steps", "ms_median_synchronised": med * 1e3,
           "ms_min_synchronised": ts[0] * 1e3, "ms_max_synchronised": ts[-1] * 1e3, "steps": steps,
           "samples_per_s": x.shape[0] / dt, "loss": float(loss.detach()),
           # forward + backward = 3 x the forward pass's FLOP (input and weight gradients each repeat its products)
           "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": 157.3, "achieved": flop / dt / 1e12,
                        "frac": flop / dt / 157.3e12, "flop_per_step": flop},
           "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}
    flat.release()
    return rec


def secondary(model, x):
    """Untimed extras for the record (never part of `value`; any failure is reported, not raised): the other BASELINE
    configurations through tools/config_bench.py and one training step of the benchmark model (forward_kld + backward +
    Adam through the autograd path).  stdout of the helpers is swallowed: the contract is ONE JSON line."""
    import contextlib
    import importlib.util
    import io
    res = {}
    try:
        spec = importlib.util.spec_from_file_location("nf_config_bench", os.path.join(ROOT, "tools", "config_bench.py"))
        cb = importlib.util.module_from_spec(spec)
        with contextlib.redirect_stdout(io.StringIO()):
            spec.loader.exec_module(cb)
            for key, fn in (("config1_realnvp", cb.c1), ("config4_glow", cb.c4), ("config5_maf", cb.c5), ("nsf_wide", cb.wide)):
                try:
                    res[key] = fn()
                except Exception as exc:   # noqa: BLE001
                    res[key] = {"error": repr(exc)[:200]}
    except Exception as exc:   # noqa: BLE001
        res["config_bench"] = {"error": repr(exc)[:200]}
    try:
        res["train_step"] = train_step_record(model, x, world=1)
    except Exception as exc:   # noqa: BLE001
        res["train_step"] = {"error": repr(exc)[:200]}
    try:
        res["other_batches"] = other_batches_record(model, x)
    except Exception as exc:   # noqa: BLE001
        res["other_batches"] = {"error": repr(exc)[:200]}
    return res


def other_batches_record(model, x):
    """log_prob of the benchmark model at smaller batches (round 6, late): a workgroup owns its rows for the whole 32-layer chain, so a
    pass costs the same for any batch one round of workgroups holds; at <= 32 768 rows nf_rqs_fused_chain switches to 128-row
    workgroups (csrc/rqs_fused_nw4.hip, bit-identical per row).  Eager launches, 20 passes each."""
    out = {"workload": "log_prob of the BASELINE configs[1] model on the first B benchmark rows (eager launches, mean of 20 passes)"}
    with torch.no_grad():
        for B in (4096, 16384, 32768):
            xb = x[:B].contiguous()
            for _ in range(3):
                model.log_prob(xb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                model.log_prob(xb)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / 20
            out["rows_%d" % B] = {"log_prob_ms": ms, "rows_per_s": B / ms * 1e3}
    return out


def train_mode(args, model, x, world, rank, dev):
    """`bench.py --gpus N --train`: W untimed + K timed training steps per rank (make_train_step: forward_kld + backward with the
    bucketed gradient all-reduce started from autograd hooks during backward + Adam on the flat parameter), barrier + synchronize on
    both sides, max over ranks; rank 0 prints ONE JSON line.  Collectives of the timed region are counted (every entry point)."""
    import torch.distributed as dist
    one_step, flat, opt = make_train_step(model, x, world)

    def barrier():
        if world > 1:
            dist.barrier()
    for _ in range(max(args.warmup, 1)):
        loss = one_step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    counter = CollectiveCounter(world > 1)
    with counter:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = one_step()
        torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # every rank ends the run with the same weights: the averaged gradient is the same on all of them
    chk = flat.param.detach().double().sum().reshape(1)
    same = True
    if world > 1:
        lo_, hi_ = chk.clone(), chk.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        same = bool((lo_ == hi_).item())
    flop = 3 * c2_flops_per_sample(layers=args.layers) * args.batch
    dt = elapsed / args.steps
    if rank == 0:
        print(json.dumps({
            "metric": "training samples/sec (forward_kld + backward + gradient all-reduce + Adam), 32-layer RQ-NSF d=64 B=65536",
            "value": world * args.batch / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "loss": float(loss.detach()), "replicas_identical_after_run": same,
            "config": {"workload": "BASELINE configs[1] model, TRAINING step: forward_kld + backward + Adam(fused) on "
                                   "dp.FlatParameters, %d rows/GPU; gradients averaged by 8 MB in-place all-reduces of the flat "
                                   "gradient buffer started during backward" % args.batch,
                       "rows_per_gpu": args.batch, "global_rows": world * args.batch, "layers": args.layers,
                       "parallelism": "dp%d" % world, "collectives_per_step": counter.coll["calls"] / max(args.steps, 1),
                       "collective_bytes_per_step": counter.coll["bytes"] / max(args.steps, 1),
                       "collectives_by_op": counter.coll["by_op"]},
            "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": 157.3, "achieved": flop / dt / 1e12,
                         "frac": flop / dt / 157.3e12, "traffic": None, "flop_per_step_per_gpu": flop,
                         "note": "3 x the forward pass's algorithmic FLOP per rank and step over the whole step's wall time"}}))
    flat.release()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="rows per GPU")
    ap.add_argument("--layers", type=int, default=LAYERS)
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the untimed extra measurements (other BASELINE configs, "
                    "training step) that rank 0 appends under \"secondary\" at N = 1")
    ap.add_argument("--cpu-rows", type=int, default=131072, help="rows of the same workload timed on the host oracle")
    ap.add_argument("--train", action="store_true", help="time the TRAINING step instead (forward_kld + backward + overlapped gradient "
                    "all-reduce + Adam per rank and step; rows/s over all ranks) and print its own JSON line: the second curve of a "
                    "multi-GPU lease (VERDICT r05 #8); the default line (the BASELINE metric) is unchanged")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher ourselves -- one rank per GPU under torch.distributed.run on
        # this node, rendezvous on 127.0.0.1 (the container hostname may not resolve), same arguments.  Under the driver's own
        # `python -m torch.distributed.run ... bench.py --gpus N` WORLD_SIZE is set and this branch is not taken.
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        if os.environ.get("NF_BENCH_PRINT_LAUNCH") == "1":      # CPU test hook: show the launch line, start nothing
            print(json.dumps({"launch": cmd}))
            return
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    import torch.distributed as dist
    import normflows_amd as nfa
    from normflows_amd import dp

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # NF_BENCH_ONE_DEVICE=1 (testing only): all ranks share cuda:0 and talk over gloo, to exercise the N > 1 code path on
    # a 1-GPU box; the real multi-GPU run uses one rank per GPU over RCCL (backend "nccl").
    one_dev = os.environ.get("NF_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if (world == 1 or one_dev) else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)
    dev = torch.device("cuda", dev_index)

    model = build_c2_model(num_layers=args.layers).to(dev)
    x = c2_inputs(args.batch, DIM, rank=rank).to(dev)
    if args.train:
        train_mode(args, model, x, world, rank, dev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    model.use_graphs(not args.no_graph)

    def barrier():
        if world > 1:
            dist.barrier()

    def step():
        """One pass of the hot path over the rank's batch: log_prob of every row, then the NLL over the GLOBAL batch --
        the path's single collective (an 8-byte all-reduce; a local reduction at N = 1)."""
        lp_ = model.log_prob(x)
        return lp_, dp.global_nll(lp_)

    with torch.no_grad():
        for _ in range(max(args.warmup, 1 if not args.no_graph else 0)):
            lp, nll_t = step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        # the data-path collectives of the timed region are COUNTED (calls, bytes) as they are issued: SURVEY.md 8e allows exactly
        # one 16-byte all-reduce ([sum log_q, n], fp64) per step; the barriers / the max-over-ranks clock below are not data path
        counter = CollectiveCounter(world > 1)
        with counter:                       # (restores the entry points also when a step raises: ADVICE r05)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                lp, nll_t = step()
            torch.cuda.synchronize()
        coll = counter.coll
        barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    nll = float(nll_t.item()) / DIM                  # from the last timed step (the path's 8-byte all-reduce)
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.batch * args.steps / elapsed

    out = {
        "metric": "samples/sec + NLL (nats/dim), 32-layer RQ-NSF d=64 B=65536",
        "value": value, "unit": "samples/s", "nll_nats_per_dim": nll,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: log_prob of %d x [CoupledRQS(d=64, 2 blocks, 128 hidden, K=8, "
                               "tail 3) + LULinearPermute(64)] + DiagGaussian, batch %d rows/GPU, N(0,I) inputs, "
                               "sigma=0.01 perturbed seeded weights" % (args.layers, args.batch),
                   "rows_per_gpu": args.batch, "global_rows": world * args.batch, "layers": args.layers,
                   "launch": "eager" if args.no_graph else "hipGraph replay", "parallelism": "dp%d" % world,
                   "collectives_per_step": coll["calls"] / max(args.steps, 1),
                   "collective_bytes_per_step": coll["bytes"] / max(args.steps, 1)},
    }
    if rank == 0:
        flops = c2_flops_per_sample(layers=args.layers)
        out["end_to_end"] = {"tflops": flops * value / 1e12 / world, "frac_of_fp32_mfma_peak":
                             flops * value / world / 157.3e12, "flop_per_sample": flops}
        if not args.no_breakdown:
            model.use_graphs(False)
            with torch.no_grad():
                bd = kernel_breakdown(model, x)
            model.use_graphs(not args.no_graph)
            out["kernel_ms"] = {k: {"avg_ms": v[0], "launches_per_pass": v[1], "layer_pairs_per_launch": v[2]}
                                for k, v in bd.items()}
            chain_ms, _, npairs = bd["rqs_fused_chain"]
            if npairs:
                # dominant kernel: nf::rqs_fused_kernel<0, true, false, 8>, ONE persistent launch over all layer pairs.
                # Algorithmic FLOPs per launch: (327 680 conditioner + 16 384 LU) FLOP per sample and layer
                # pair x rows x pairs; MFMA-bound (exact-fp32 MFMA, 157.3 TFLOP/s peak).  chain_ms also contains the
                # torch.zeros fill of log_q (one 256 KB memset).
                fl = c2_flops_per_sample(layers=npairs) * args.batch
                ach = fl / (chain_ms * 1e-3) / 1e12
                tr, tr_src = pmc_traffic()
                out["roofline"] = {"kernel": "nf::rqs_fused_kernel<0, true, false, 8>", "bound": "mfma", "achieved": ach,
                                   "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3, "traffic": tr,
                                   "traffic_source": tr_src,
                                   "flop_per_launch": fl, "avg_launch_ms": chain_ms, "layer_pairs_per_launch": npairs,
                                   "hbm_algorithmic_bytes_per_launch": (2 * DIM * 4 + 8) * args.batch}
                try:        # extra, not part of the contract's fields: the clock the MFMA pipe actually runs at under load
                    from normflows_amd import ops as _ops
                    # a 5 ms burst of back-to-back fp32 MFMAs on every SIMD (the length of one chain launch) right after the
                    # timed region, i.e. on a warm chip: ~2.4 GHz, the clock the guide's 157.3 TFLOP/s assumes (from idle the
                    # same burst runs at 2.2 GHz: isolated micro-benchmarks see up to 10 % less MFMA rate than a sustained run)
                    out["roofline"]["shader_clock_mhz_mfma_probe_5ms"] = _ops.mfma_clock_mhz(dev, 6000)
                except Exception as exc:   # noqa: BLE001
                    out["roofline"]["shader_clock_mhz_mfma_probe_5ms"] = repr(exc)[:100]
        if not args.no_breakdown:
            # secondary (SURVEY.md 8d reports both directions): generative pass = every layer's forward + log_q
            g = torch.Generator().manual_seed(4321)
            eps = torch.randn(args.batch, DIM, generator=g).to(dev)
            with torch.no_grad():
                for _ in range(2):
                    model.sample_from_noise(eps)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(max(args.steps // 2, 3)):
                    xs, lq = model.sample_from_noise(eps)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / max(args.steps // 2, 3)
            out["sample_direction"] = {"value": args.batch / dt, "unit": "samples/s", "ms_per_step": 1e3 * dt}
        if not args.no_breakdown:
            # secondary: the same pass with the fused kernel's GEMMs on the bf16 matrix pipe by error-compensated
            # splitting (fp32 = hi + mid + lo bf16, six products, fp32 accumulation; csrc/rqs_fused_x3.hip).  Same
            # parity tests as the exact-fp32 kernel; reported separately, `value` above is the exact-fp32 MFMA path.
            nfa.config.set_fused_gemm("bf16x3")
            model.use_graphs(False)
            model.use_graphs(not args.no_graph)
            with torch.no_grad():
                for _ in range(2):
                    lp3 = model.log_prob(x)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    lp3 = model.log_prob(x)
                torch.cuda.synchronize()
                dt3 = (time.perf_counter() - t1) / args.steps
            rel = float(((lp3 - lp).abs() / lp.abs().clamp_min(1.0)).max())
            # roofline of ITS dominant kernel: nf::rqs_fused_x3_kernel<0, true>, ONE persistent launch over all layer pairs
            # (nf_rqs_fused_x3_chain); every fp32 product is six bf16 MFMA products (hi.hi + hi.mid + mid.hi + hi.lo + lo.hi +
            # mid.mid), priced against the dense bf16 peak
            model.use_graphs(False)
            with torch.no_grad():
                bd3 = kernel_breakdown(model, x)      # same instrumented pass as the exact kernel's, config still bf16x3
            model.use_graphs(not args.no_graph)
            launches3 = (args.layers + 63) // 64                            # chains of up to 64 pairs
            fl3 = 6 * c2_flops_per_sample(layers=args.layers) * args.batch / launches3     # executed bf16 FLOP per launch
            launch_ms = bd3["rqs_fused_chain"][0] / launches3               # HIP events around the chain launch(es)
            out["bf16x3_split_gemm"] = {"value": args.batch / dt3, "unit": "samples/s", "ms_per_step": 1e3 * dt3,
                                        "nll_nats_per_dim": float(-lp3.mean()) / DIM,
                                        "max_rel_diff_log_prob_vs_exact_f32": rel,
                                        "executed_bf16_tflops": 6 * c2_flops_per_sample(layers=args.layers) * args.batch / dt3 / 1e12,
                                        "frac_of_bf16_mfma_peak_2500": 6 * c2_flops_per_sample(layers=args.layers) * args.batch / dt3 / 2.5e15,
                                        "roofline": {"kernel": "nf::rqs_fused_x3_kernel<0, true>", "bound": "mfma", "unit": "TFLOP/s",
                                                     "achieved": fl3 / (launch_ms * 1e-3) / 1e12, "peak": 2500.0,
                                                     "frac": fl3 / (launch_ms * 1e-3) / 2.5e15, "traffic": None,
                                                     "flop_per_launch_executed_bf16": fl3, "avg_launch_ms": launch_ms,
                                                     "launches_per_pass": launches3, "layer_pairs_per_launch": min(args.layers, 64),
                                                     "note": "avg_launch_ms = HIP events around the chain launch on its stream; "
                                                             "algorithmic fp32 FLOP = executed / 6"}}
            nfa.config.set_fused_gemm("f32")
            model.use_graphs(False)
            model.use_graphs(not args.no_graph)
        if not args.no_cpu_baseline:     # rank 0 only, untimed (at N > 1 the other ranks wait at the closing barrier)
            out["cpu_baseline"] = cpu_baseline(model, args.cpu_rows, lp.cpu().numpy(),
                                               lp3.cpu().numpy() if not args.no_breakdown else None)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            ref_leg = reference_cpu_leg()
            if ref_leg is not None:
                out["cpu_baseline"]["reference"] = ref_leg
                out["gpu_over_cpu_reference"] = value / ref_leg["value"]
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary(model, x)
        print(json.dumps(out))
    if world > 1:
        barrier()   # rank 0's untimed extras are done: every rank tears the communicator down together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
