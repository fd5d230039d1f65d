#!/usr/bin/env python
"""bench.py -- throughput of the FGSM-family hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
        N > 1 without a launcher: this process re-launches itself as N ranks (torch.distributed.run, one per GPU,
        backend nccl = RCCL, rendezvous on 127.0.0.1) and fails loudly if fewer than N HIP devices are visible;
        under torch.distributed.run (RANK / WORLD_SIZE set): it is one of those ranks.

Workload (BASELINE.json configs[1]): MI-FGSM on ResNet-50, eps=16/255, alpha=1.6/255, K=10 iterations,
synthetic 3x224x224 images.  One "step" = one batch of 125 images (1000/8: the per-GPU shard of the 1000-image
set; 8 steps = the whole set) through ``attacker(images, labels)`` = 10 x (surrogate forward + input-gradient
backward + fused HIP update).  Host-side arrangement of the surrogate (all reported in ``config``, all
switchable): eval-mode BatchNorm folded into the convolutions, NHWC memory format (profiles/r01/backbone_probe.jsonl:
+46% over plain NCHW at this batch), and -- with both -- the fused execution of backbones/fused.py (same MIOpen
convolutions, the memory-bound passes between them fused, the stem's input gradient on the fp32-MFMA kernel:
TA_FUSED_GLUE / TA_STEM_KERNEL = 0 switch them off) and, site by site where it measures faster, convolution + glue pass as ONE
composable_kernel convolution with the pass as epilogue (libta_ck.so, ``--ck-epilogue 0`` switches it off);
``--batch 32 --fold-bn 0 --channels-last 0`` is the reference's literal setup, which every line also reports as ``config.literal``.
Inputs are resident in HBM before the timed region; the surrogate is the ResNet-50 architecture with seeded
random weights (no checkpoints offline); arithmetic is fp32 throughout, as in the reference.

Multi-GPU: the 1000-image job shards by whole batches, no data-path collective: every rank runs
its own K steps ("weak" scaling); value = images of all ranks / max-over-ranks time.  ``--attack ens --model a,b,c,d``
on N = k*4 GPUs puts one surrogate per rank of a 4-rank model group (configs[4]): the two all-reduces of the ensemble
path (logits forward, input gradient backward) run over RCCL, images shard over the k groups.

One JSON line on rank 0, with
  roofline      the fused momentum-sign-project update (ta_mi_update_std / ta_mi_update).  ``achieved`` / ``frac``: the bytes
                every launch's kernel instantiation REQUESTS -- 4 B per fp32 operand read or written, 1 B per element of the
                image when it comes from the byte source (images are PNG-decoded bytes): 21 B/element in the steady state of
                the default loop (read gy, m, delta, the image byte; write m, delta), 17 on the first iteration (no momentum
                yet) -- summed over the launches / summed launch durations, measured with HIP events on the launch stream
                inside the timed region: events bound to the kernels' own dispatch packets (ta_timing_begin /
                ta_timing_end; ``roofline.clock``), the hipEventRecord-marker clock of the same launches beside it
                (``roofline.marker_clock``); peak 8 TB/s (MI355X_MICROARCH.md; 6.29 TB/s is what a float4 copy reaches).
                ``frac_at_24B_contract``: the steady-state launches at SURVEY 8(d)'s algorithmic 24 B/element (read g, m,
                delta, x; write m, delta -- what the launch does, since round 5 it stores no x + delta);
                ``frac_algorithmic``: every operand priced as fp32; ``traffic``: the committed PMC passes
                (profiles/pmc_update_kernel.json) priced per launch.
  config.literal  the SAME attack timed in the reference-literal arrangement for a few steps (``--literal-steps``): batches of
                32, NCHW, separate BatchNorm, nn.Module execution, MIOpen immediate mode -- what main.py runs with no
                environment set; ``images_per_s``, the update's ``update_frac`` (executed bytes; a sum-only pass precedes each
                update there: ``k1_passes``)
  cpu_baseline  the oracle (oracle/fgsm_oracle.py = the reference's ATen CPU arithmetic) on the host cores: 32 images (the
                reference's batch), all K=10 iterations, at the best of a thread sweep from 8 to every hardware thread
                (``cores`` = physical cores, ``threads_used``, ``thread_sweep_images_per_s``); kind "reference" where
                /root/reference exists (the build container), "port" on the GPU box.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (6290 GB/s measured float4 copy)
BYTES_PER_ELEM = 24            # fused update: r g,m,d,x  w m,d


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=8)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--attack", default="mifgsm")
    p.add_argument("--model", default="resnet50")
    p.add_argument("--batch", type=int, default=125,
                   help="images per step; MI-FGSM treats images independently, 125 = the per-GPU shard of the "
                        "1000-image set on 8 GPUs (the reference CLI default is 32)")
    p.add_argument("--fold-bn", type=int, default=1,
                   help="fold the surrogate's eval-mode BatchNorm into its convolutions (algebraically exact)")
    p.add_argument("--channels-last", type=int, default=1, help="run the surrogate in NHWC memory format")
    p.add_argument("--ck-epilogue", type=int, default=1,
                   help="fused ResNet path: convolution + glue pass as ONE composable_kernel convolution with the pass as epilogue "
                        "(libta_ck.so) wherever that measures faster than MIOpen's convolution + the glue kernel (TA_CK_EPILOGUE)")
    p.add_argument("--cpu-images", type=int, default=32, help="images of the CPU-baseline sample (0 = skip); 32 = the reference's batch")
    p.add_argument("--kernel-sweep", type=int, default=1, help="also time the update kernel stand-alone")
    p.add_argument("--literal-steps", type=int, default=3,
                   help="also time the reference-literal arrangement (batch 32, NCHW, separate BatchNorm, MIOpen immediate "
                        "mode) for this many steps -> config.literal (0 = skip)")
    p.add_argument("--kernel-times", type=int, default=0,
                   help="time every HIP kernel call of the loop with events (config.kernels); for the transform attacks")
    # the four flags below exist for tests/test_bench_ranks.py: the multi-rank reporting path (barrier, MAX all-reduce of
    # the time, all-gather of the rates, the sharded-ensemble layout) on gloo / CPU tensors with the kernels' host build
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for the CPU tests)")
    p.add_argument("--device", default="cuda", help="cuda (= HIP); cpu only under the tests' host stand-in of the kernels")
    p.add_argument("--share-device", type=int, default=0,
                   help="diagnostic: every rank drives device 0 (with --backend gloo; the one-GPU cold-start probe of staged_warmup)")
    p.add_argument("--image-size", type=int, default=224)
    p.add_argument("--classes", type=int, default=1000)
    return p.parse_args(argv)


def synthetic_batch(n, seed, size=224, classes=1000):
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, 256, (n, 3, size, size), generator=g, dtype=torch.uint8).float() / 255
    y = torch.randint(0, classes, (n,), generator=torch.Generator().manual_seed(seed + 1))
    return x, y


def kernel_sweep(sizes=(32, 125, 250), reps=30):
    """Stand-alone timing of the fused update (K1 + K2: no producer has left |g| sums) at several batch sizes; operands
    rotate through 4 buffer sets, ~0.3-2.4 GB, so the 256 MiB Infinity Cache cannot hold them."""
    from transferattack_amd import _hip
    out = {}
    e = 3 * 224 * 224
    for n in sizes:
        sets = []
        for k in range(4):
            g = torch.randn(n, 3, 224, 224, device="cuda") * 1e-4
            sets.append((g, torch.randn_like(g), torch.zeros_like(g), torch.rand_like(g)))
        for i in range(5):
            g, m, d, x = sets[i % 4]
            _hip.mi_update(g, m, m, d, x, 1.0, 1.6 / 255, 16 / 255)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        start.record()
        for i in range(reps):
            g, m, d, x = sets[i % 4]
            _hip.mi_update(g, m, m, d, x, 1.0, 1.6 / 255, 16 / 255)
        end.record()
        torch.cuda.synchronize()
        us = start.elapsed_time(end) * 1e3 / reps
        out["n%d_k1_k2" % n] = {"us": round(us, 2), "GBps": round(BYTES_PER_ELEM * e * n / us / 1e3, 1)}
        # the steady-state launch of the loop (|g| sums handed over, x + delta written: 28 B/element algorithmic) by the
        # dispatch clock, with the fp32 image operand and with the byte source, same cold operands
        try:
            xa = torch.empty_like(sets[0][0])
            for tag, byte_valued in (("fp32_source", False), ("byte_source", True)):
                srcs = []
                for k in range(4):
                    if byte_valued:             # float(byte) / 255 with the IEEE quotient's bits, built on the device
                        xb = (torch.randint(0, 256, sets[k][3].shape, device="cuda", dtype=torch.uint8).double() / 255).float()
                        sets[k] = sets[k][:3] + (xb.contiguous(),)
                        srcs.append(_hip.u8_source_probe(sets[k][3]))
                    else:
                        srcs.append(None)
                _hip.timing_begin(reps + 8)
                for i in range(reps):
                    g, m, d, x = sets[i % 4]
                    _hip.abs_sum_partials(g)
                    _hip.mi_update(g, m, m, d, x, 1.0, 1.6 / 255, 16 / 255, x_adv=xa, data_u8=srcs[i % 4])
                torch.cuda.synchronize()
                ms = _hip.timing_end()[5:]                                  # the first launches warm the instruction cache
                us = 1e3 * sum(ms) / len(ms)
                taken = (not byte_valued) or all(int(s_[1].item()) == 0 for s_ in srcs)
                out["n%d_k2_steady_%s" % (n, tag)] = {"us": round(us, 2), "GBps_at_28B": round(28 * e * n / us / 1e3, 1),
                                                      "frac_of_8TBps": round(28 * e * n / us / 1e3 / HBM_PEAK_GBS, 4),
                                                      "bytes_read_by_the_kernel": bool(taken and byte_valued)}
            # round 5: the std form (gy / std[c] inline, no x_adv store), sums handed over as the stem kernel hands them
            std = torch.tensor([0.229, 0.224, 0.225], device="cuda")
            _hip.timing_begin(reps + 8)
            for i in range(reps):
                g, m, d, x = sets[i % 4]
                _hip.abs_sum_partials_std(g, std)
                _hip.mi_update(g, m, m, d, x, 1.0, 1.6 / 255, 16 / 255, data_u8=srcs[i % 4], std=std)
            torch.cuda.synchronize()
            ms = _hip.timing_end()[5:]
            us = 1e3 * sum(ms) / len(ms)
            out["n%d_k2_std_steady_byte_source" % n] = {"us": round(us, 2), "executed_GBps_at_21B": round(21 * e * n / us / 1e3, 1),
                                                        "frac_executed": round(21 * e * n / us / 1e3 / HBM_PEAK_GBS, 4),
                                                        "frac_at_24B_contract": round(24 * e * n / us / 1e3 / HBM_PEAK_GBS, 4)}
        except Exception as exc:  # noqa: BLE001  -- a diagnostic: never at the expense of the line
            out["n%d_k2_steady_error" % n] = repr(exc)[:160]
            try:
                _hip.timing_end()                                           # leave no timing session armed behind
            except Exception:  # noqa: BLE001
                pass
        del sets
    return out


# algorithmic bytes of one call of a binding function, from its arguments
_NB = lambda t: 4 * t.numel()                                                   # noqa: E731
KERNEL_BYTES = {
    "normalize_fwd": lambda x, y, *a: _NB(x) + _NB(y),
    "normalize_bwd": lambda gy, gx, *a, **k: _NB(gy) * (3 if k.get("variance") is not None else 2),
    "vmi_neighbor_normalized": lambda data, delta, out, *a, **k: 3 * _NB(data),
    "normalize_bwd_accumulate": lambda gy, acc, std, first: _NB(gy) * (2 if first else 3),
    "momentum": lambda g, m_in, m_out, *a, **k: _NB(g) * (2 if m_in is None else 3),
    "update_delta_linf": lambda d, x, m, *a, **k: 4 * _NB(d),
    "depthwise_conv2d_same": lambda inp, out, w: _NB(inp) + _NB(out),
    "sum_members": lambda grads, gx: _NB(gx) * (len(grads) + 1),
    "dim_fwd": lambda x, y, *a: _NB(x) + _NB(y),
    "dim_bwd": lambda gy, gx, *a: _NB(gy) + _NB(gx),
    "scale_copies_fwd": lambda x, y, *a: _NB(x) + _NB(y),
    "scale_copies_bwd": lambda gy, gx, *a: _NB(gy) + _NB(gx),
    "sum_copies_bwd": lambda gy, gx, *a: _NB(gy) + _NB(gx),
    "admix_fwd": lambda x, perm, y, num_admix, *a: _NB(x) * (1 + num_admix) + _NB(y),
    "admix_bwd": lambda gy, gx, *a: _NB(gy) + _NB(gx),
    "sia_fwd": lambda x, plan, y, *a, **k: _NB(x) + _NB(y),
    "sia_bwd": lambda gy, plan, x, gx, *a, **k: _NB(gy) + 2 * _NB(x),
    "bsr_fwd": lambda x, plan, y, *a: _NB(x) + _NB(y),
    "bsr_bwd": lambda gy, plan, gx, *a: _NB(gy) + _NB(gx),
    "dct_pair": lambda inp, add, mul, out, *a: _NB(inp) * (2 + (add is not None) + (mul is not None)),
    "vmi_neighbor": lambda data, delta, out, *a, **k: 3 * _NB(data),
    "grad_accumulate": lambda acc, grad, first: _NB(acc) * (2 if first else 3),
    "variance_finalize": lambda acc, cur, out, *a: 3 * _NB(acc),
    "axpy": lambda x, m, coeff, out: 3 * _NB(x),
}


def instrument_kernels(module, make_event):
    """Wrap the binding functions named in KERNEL_BYTES with event timing (bench-local: the product is not touched).
    Returns (records, restore): records[name] = [(start, end, bytes)], restore() puts the originals back."""
    records, originals = {}, {}

    def wrap(name, fn):
        def timed(*args, **kwargs):
            start, end = make_event(), make_event()
            start.record()
            out = fn(*args, **kwargs)
            end.record()
            records.setdefault(name, []).append((start, end, KERNEL_BYTES[name](*args, **kwargs)))
            return out
        return timed

    for name in KERNEL_BYTES:
        originals[name] = getattr(module, name)
        setattr(module, name, wrap(name, originals[name]))

    def restore():
        for name, fn in originals.items():
            setattr(module, name, fn)
    return records, restore


def summarise_kernels(records):
    out = {}
    for name, rows in sorted(records.items()):
        us = [s.elapsed_time(e) * 1e3 for s, e, _ in rows]
        mean_us, nbytes = sum(us) / len(us), sum(b for _, _, b in rows) / len(rows)
        out[name] = {"launches": len(rows), "mean_us": round(mean_us, 2), "algorithmic_bytes": int(nbytes),
                     "GBps": round(nbytes / mean_us / 1e3, 1)}
    return out


def physical_cores():
    """physical cores of this host (unique (socket, core) pairs of /proc/cpuinfo); hardware threads if that fails"""
    try:
        pairs, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((phys, line.split(":")[1].strip()))
        return len(pairs) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_baseline(args):
    """The reference's CPU path on this host's cores: ``cpu_images`` images (default 32, the reference's batch,
    main.py:15) through ALL K=10 iterations (one surrogate forward/backward + the 13-kernel update stack each), at the
    best of a thread sweep from 8 towards every hardware thread of the host (one iteration per count; the sweep stops at the
    first count whose warm-up iteration takes over 3x the best count's -- beyond the optimum ATen's intra-op parallelism only
    loses on a 32-image batch: the complete sweep to 256 threads is on record in profiles/r04/bench_default_b125_r4e.json).
    A slow host (one iteration > 6 s) gets a proportionally shorter timed run, scaled to K=10 and said so in ``sample``.
    kind "reference": the reference's OWN ``Attack.forward`` (imported from /root/reference through oracle/ref_shim.py --
    only where that tree exists, i.e. the build container); kind "port": the oracle (oracle/fgsm_oracle.py, the same ATen
    ops in the same order) -- what runs on the GPU box."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fgsm_oracle as O
    import ref_shim
    from transferattack_amd import backbones
    hw_threads = os.cpu_count() or 1
    cores = physical_cores()
    model = backbones.create(args.model, seed=0, verbose=False)
    x, y = synthetic_batch(args.cpu_images, 0)
    kind = "port"
    run = lambda epoch: O.run_attack(args.attack, model, x, y, epoch=epoch)          # noqa: E731
    if ref_shim.reference_available():
        try:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):
                ref_attack = ref_shim.make_reference_attack(args.attack, model)

            def run(epoch):                                                          # noqa: F811
                ref_attack.epoch = epoch
                return ref_attack(x, y)
            kind = "reference"
        except Exception as exc:  # noqa: BLE001
            print("reference attack class unavailable (%r): timing the oracle port" % (exc,), file=sys.stderr)
    # thread sweep: torch's default (one thread per hardware thread) is rarely the fastest on a 2-socket host
    sweep, best = {}, (float("inf"), torch.get_num_threads())
    counts = sorted({min(t, hw_threads) for t in (8, 16, 32, 64, 128, cores, hw_threads)})
    budget_s, t_sweep = 45.0, time.time()
    for th in counts:
        if time.time() - t_sweep > budget_s:
            break
        torch.set_num_threads(th)
        t0 = time.time()
        run(1)                                                         # warm-up at this count
        if best[0] < float("inf") and time.time() - t0 > 3 * best[0]:
            # more threads only make it slower from here (r4b / r4e on 2 x EPYC 9575F: 16 threads 2.6 images/s, 64: 1.2, 128:
            # 0.5, 256: 0.04 -- the last probe alone took three minutes); the full sweep is on record in profiles/r04
            sweep[th] = "skipped: warm-up %.1f s, over 3x the best count's iteration" % (time.time() - t0)
            break
        t0 = time.time()
        run(1)
        dt1 = time.time() - t0
        sweep[th] = round(args.cpu_images / (dt1 * 10), 3)             # images/s if all 10 iterations ran at this pace
        best = min(best, (dt1, th))
    threads = best[1]
    torch.set_num_threads(threads)
    cpu_iters = 10 if best[0] <= 6.0 else max(1, int(60.0 / best[0]))
    t0 = time.time()
    run(cpu_iters)
    dt = (time.time() - t0) / cpu_iters * 10
    # update stack alone (get_momentum + update_delta), the reference's op string, N = 32
    n = 32
    g = torch.randn(n, 3, 224, 224)
    m, d, xx = torch.randn_like(g), torch.zeros_like(g), torch.rand_like(g)
    for _ in range(2):
        mm = O.momentum_step(g, m, 1.0)
        O.delta_step(d, xx, mm, 1.6 / 255, 16 / 255)
    t1 = time.time()
    for _ in range(5):
        mm = O.momentum_step(g, m, 1.0)
        O.delta_step(d, xx, mm, 1.6 / 255, 16 / 255)
    upd_ms = (time.time() - t1) / 5 * 1e3
    return {"value": round(args.cpu_images / dt, 4), "unit": "images/s", "cores": cores, "threads_used": threads,
            "hw_threads": hw_threads, "kind": kind, "thread_sweep_images_per_s": sweep,
            "sample": "%d synthetic images (the reference's batch), %s on %s, %s, torch CPU with %d threads -- the best of the "
                      "sweep -- on a host with %d physical cores / %d hardware threads (%s)" % (
                          args.cpu_images, args.attack, args.model,
                          "all K=10 iterations timed" if cpu_iters == 10 else
                          "%d of the K=10 iterations timed and scaled to 10" % cpu_iters, threads, cores, hw_threads,
                          "the reference's own Attack.forward via oracle/ref_shim.py" if kind == "reference"
                          else "oracle/fgsm_oracle.py; /root/reference is not present on this host"),
            "update_stack_ms_n32": round(upd_ms, 3),
            "update_stack_GBps_n32": round(BYTES_PER_ELEM * 150528 * n / upd_ms / 1e6, 2)}


def launch_ranks(args):
    """``--gpus N`` without a launcher: start N ranks of this script (one process per GPU) and wait for them."""
    import socket
    import subprocess
    visible = torch.cuda.device_count()
    if visible < args.gpus and not args.share_device:
        sys.exit("bench.py --gpus %d: only %d HIP device(s) visible on this node" % (args.gpus, visible))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def open_world(args):
    """(rank, world, local rank) of this process; joins the process group the launcher described (one rank per GPU)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d rank(s); reporting what actually runs" % (args.gpus, world),
              file=sys.stderr)
    if args.share_device:
        local = 0
        os.environ["LOCAL_RANK"] = "0"              # transferattack_amd.utils.default_device reads it
    if args.device == "cuda":
        torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            if args.device == "cuda" and args.backend == "nccl":
                dist.init_process_group(args.backend, device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(args.backend)
    return rank, world, local


def build_attacker(args, world):
    """-> (attacker, index of this rank's image shard, number of image shards, description of the layout).  A list of
    surrogates on a world that is a multiple of its length: one member per rank (transferattack_amd.dist.sharded_attack)."""
    import contextlib
    import transferattack_amd as ta
    from transferattack_amd import dist as tadist
    rank = int(os.environ.get("RANK", "0"))
    model_name = args.model.split(",") if "," in args.model else args.model      # list -> EnsembleModel, as main.py:39-40
    shard_rank, shard_world, layout = rank, world, "image-shard x%d, no collective" % world
    with contextlib.redirect_stdout(sys.stderr):              # stdout carries exactly one line: the JSON result
        cls = ta.load_attack_class(args.attack)
        if isinstance(model_name, list) and world > 1 and world % len(model_name) == 0:
            attacker, member, shard_rank, shard_world = tadist.sharded_attack(cls, args.attack, model_name, world)
            layout = "%d image shard(s) x %d model ranks (one surrogate per rank; %s all-reduce of logits and input " \
                     "gradients)" % (shard_world, len(model_name), "RCCL" if args.backend == "nccl" else args.backend)
        else:
            attacker = cls(model_name=model_name)
    return attacker, shard_rank, shard_world, layout


def device_sync(args):
    if args.device == "cuda":
        torch.cuda.synchronize()


def staged_warmup(step, args, world, rank, independent=True):
    """The W untimed warm-up steps, rank 0 FIRST.  bench.py runs MIOpen in find mode (``cudnn.benchmark``): on a fresh node the
    first process spends minutes building and timing every applicable solver of every convolution (285-333 s for ResNet-50 at
    batch 125, profiles/r05/cold_start_main_1000png_r5a.jsonl) and leaves the result in MIOpen's user find-db and kernel
    cache -- files shared by all ranks of the node.  N ranks entering that find at once would do the same work N times,
    contending for the same files and for the host's cores (every solver is a clang job).  So ranks 1..N-1 wait on the host
    (a gloo barrier: no RCCL kernel spinning on their GPUs, no collective watchdog) until rank 0 has warmed up, then warm up
    together against the populated databases (find-db hits, kernels loaded from the cache).  Wall ~ cold(1) + warm(1) instead
    of N contending cold starts.  ``TA_BENCH_STAGED_WARMUP=0`` lets all ranks start together (the comparison run of
    profiles/r06).  -> seconds this rank spent in (waiting, its own warm-up)."""
    # (``independent`` False: the ranks' steps exchange data -- the sharded ensemble's all-reduces -- so no rank can step alone)
    staged = independent and world > 1 and args.warmup > 0 and os.environ.get("TA_BENCH_STAGED_WARMUP", "1") != "0"
    waited = 0.0
    group = None
    if staged:
        import datetime
        import torch.distributed as dist
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
            # one node, rendezvous on the loopback: gloo would otherwise pick its interface by resolving the host NAME, which
            # a container may not be able to do
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        group = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=2))
        if rank != 0:
            t0 = time.perf_counter()
            dist.barrier(group=group)             # released when rank 0 arrives, i.e. after its warm-up
            waited = time.perf_counter() - t0
    t0 = time.perf_counter()
    for i in range(args.warmup):
        step(i)
    device_sync(args)
    warm = time.perf_counter() - t0
    if staged and rank == 0:
        import torch.distributed as dist
        dist.barrier(group=group)
    return waited, warm, staged


def timed_region(step, args, world, before=None, rank=0, independent=True):
    """W untimed warm-up steps (rank 0 first: ``staged_warmup``), then EXACTLY K steps bracketed by barrier + device
    synchronisation on both sides.
    -> (seconds between the two brackets on this rank, seconds until this rank's own last step finished,
        (seconds waited for rank 0's warm-up, seconds of this rank's own warm-up, was the warm-up staged))."""
    import torch.distributed as dist
    startup = staged_warmup(step, args, world, rank, independent)
    if world > 1:
        dist.barrier()
    device_sync(args)
    if before is not None:
        before()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    device_sync(args)
    mine = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    device_sync(args)
    return time.perf_counter() - t0, mine, startup


def over_ranks(dt, rate, world, dev):
    """MAX of the bracketed time over the ranks and every rank's own images/s (all-gather), as the contract asks.
    -> (max seconds, [rate of rank 0, ...], world size the collective saw, backend name)"""
    if world == 1:
        return dt, [rate], 1, "none (single process)"
    import torch.distributed as dist
    if dist.get_backend() == "gloo":
        dev = "cpu"                                 # (the CPU tests, and the shared-device cold-start probe)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    rates = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(rates, torch.tensor([rate], device=dev, dtype=torch.float64))
    return float(tmax.item()), [round(float(r.item()), 2) for r in rates], dist.get_world_size(), dist.get_backend()


def roofline(args, sink, dispatch_ms, timing_note, _hip, byte_source_taken=False):
    """The fused update's launches of the timed region -> the ``roofline`` object (module docstring).

    ``achieved`` / ``frac`` price every launch with the bytes its kernel instantiation REQUESTS (round 5; the PMC passes
    agree to 0.1 %): 4 B per fp32 operand it reads or writes, 1 B per element of the image when the byte source is taken.
    ``frac_at_24B_contract`` prices the steady-state launches with SURVEY 8(d)'s algorithmic figure for the update -- 24 B
    per element: read g, m, delta, x, write m, delta -- which is exactly what a steady-state launch does now that it stores
    no x + delta (executed: 21, the image operand being a byte)."""
    # two clocks over the same launches: hipEventRecord markers before / after each call (they include the ~3 us the
    # command processor spends between a marker and a dependent kernel) and events bound to the kernels' own dispatch
    # packets (begin of the call's first kernel -> end of the update kernel: what rocprofv3 --kernel-trace reports).
    # The roofline figure uses the dispatch clock when it is sane, and always reports the marker clock beside it.
    marker_us = [rec[0].elapsed_time(rec[1]) * 1e3 for rec in sink]
    durs_us, clock = marker_us, "hipEventRecord markers around each call"
    if len(dispatch_ms) == len(sink) and all(0.0 < 1e3 * d <= m + 1.0 for d, m in zip(dispatch_ms, marker_us)):
        durs_us, clock = [1e3 * d for d in dispatch_ms], "HIP events bound to the kernels' dispatch packets"
    algorithmic = [rec[2] * rec[3] * rec[4] for rec in sink]                    # 4 B per operand moved, the image as fp32
    u8 = [bool(rec[5]) and byte_source_taken for rec in sink]
    executed = [b - 3 * rec[2] * rec[3] * int(t) for b, rec, t in zip(algorithmic, sink, u8)]
    contract = [rec[2] * rec[3] * BYTES_PER_ELEM for rec in sink]                 # 24 B/element, nothing else
    steady = [i for i, b in enumerate(algorithmic) if b == max(algorithmic)]
    # a sum pass of its own inside a call re-reads g: 4 B/element more are requested by exactly the launches that ran one
    # (recorded per launch by _hip.mi_update; records of older shape fall back to the process-wide counter)
    ran_k1 = [bool(rec[7]) if len(rec) > 7 else _hip.stats["k1_passes"] > 0 for rec in sink]
    k1 = sum(ran_k1)
    executed = [b + 4.0 * rec[2] * rec[3] * int(r) for b, rec, r in zip(executed, sink, ran_k1)]
    total_us = sum(durs_us)
    achieved = sum(executed) / total_us / 1e3                                   # GB/s over the launches of the timed region
    std_form = sum(1 for rec in sink if len(rec) > 6 and rec[6])
    pmc = committed_pmc_traffic(sink, _hip)
    st_us = sum(durs_us[i] for i in steady)
    return {"bound": "hbm",
            # what ``achieved`` / ``frac`` price: "executed-bytes/2" = requested bytes incl. the byte source (round 5) with the
            # sum pass billed to the launches that ran one (round 6); rounds 1-4 priced algorithmic bytes (``frac_algorithmic``)
            "pricing": "executed-bytes/2",
            "kernel": ("ta_mi_update_std" if std_form == len(sink) else "ta_mi_update") + (
                " (mi_update_kernel; |g| tile sums left by the kernel that produced g)" if k1 == 0 else
                " (abs_sum_partials_kernel + mi_update_kernel)"),
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "bytes": "executed: what the kernel instantiation of each launch requests (4 B per fp32 operand, 1 B per element "
                     "of the image from the byte source)",
            # NOT measured by this run (rocprofv3 cannot count from inside bench.py): the committed PMC passes over this
            # kernel (FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 correction), priced per launch; the source is named
            "traffic": pmc["bytes_per_launch"] if pmc else None,
            "traffic_source": pmc["source"] if pmc else None,
            "clock": clock if timing_note is None else clock + " (dispatch clock unavailable: %s)" % timing_note,
            "marker_clock": {"mean_us": round(sum(marker_us) / len(marker_us), 2),
                             "frac": round(sum(executed) / sum(marker_us) / 1e3 / HBM_PEAK_GBS, 4)},
            "launches": len(durs_us), "std_form_launches": std_form, "byte_source_launches": sum(u8),
            "mean_us": round(total_us / len(durs_us), 2), "min_us": round(min(durs_us), 2),
            "executed_bytes_per_launch": int(sum(executed) / len(executed)),
            "algorithmic_bytes_per_launch": int(sum(algorithmic) / len(algorithmic)),
            "steady_state_launch": {"executed_bytes": int(sum(executed[i] for i in steady) / len(steady)),
                                    "algorithmic_bytes": int(max(algorithmic)),
                                    "mean_us": round(st_us / len(steady), 2),
                                    "executed_GBps": round(sum(executed[i] for i in steady) / st_us / 1e3, 1)},
            "k1_pass_skipped_launches": len(sink) - k1, "k1_passes": k1,
            "frac_of_measured_copy_peak_6290": round(achieved / 6290.0, 4),
            # SURVEY 8(d)'s contract: 24 B per element and nothing else over the steady-state launches
            "frac_at_24B_contract": round(sum(contract[i] for i in steady) / st_us / 1e3 / HBM_PEAK_GBS, 4),
            # every operand as fp32 (what the same launches would move without the byte source)
            "frac_algorithmic": round(sum(algorithmic) / total_us / 1e3 / HBM_PEAK_GBS, 4),
            # kept for readers of earlier rounds' lines: identical to achieved / frac since round 5
            "executed": {"byte_source_launches": sum(u8), "bytes_per_launch": int(sum(executed) / len(executed)),
                         "GBps": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "frac_of_measured_copy_peak_6290": round(achieved / 6290.0, 4)}}


def byte_source_taken(batches):
    """did the probe of the bench's image batches find them byte-valued (device flag 0)?  Read AFTER the timed region."""
    from transferattack_amd import _hip
    if os.environ.get("TA_U8_SOURCE", "1") == "0" or not batches[0][0].is_cuda:
        return False
    try:
        return all(int(_hip.u8_source_probe(x.contiguous())[1].item()) == 0 for x, _ in batches)
    except (_hip.HipExtensionError, ValueError):
        return False


def committed_pmc_traffic(sink, _hip):
    """HBM bytes per launch priced with the COMMITTED rocprofv3 PMC passes over the shipped kernel
    (profiles/pmc_update_kernel.json, tools/gpu_check.sh pmc, N = 125): each launch of the timed region with the counters
    of its own kernel instantiation (steady state / first iteration / decay 0 move different operands).  A constant read
    from a file, labelled as such -- not a measurement of this run."""
    pmc_path = os.path.join(ROOT, "profiles", "pmc_update_kernel.json")
    if not os.path.isfile(pmc_path):
        return None
    kernels = json.load(open(pmc_path))["kernels"]
    per_shape = {}
    for name, c in kernels.items():
        if "mi_update_kernel<" not in name:
            continue
        flags = [f.strip() for f in name[name.index("<") + 1:name.rindex(">")].split(",")]
        # <VEC, BLOCK, SLOTS, NT, HAS_V, HAS_MIN, HAS_MOUT, HAS_XADV[, X_U8[, G_STD]]>
        key = tuple(f == "true" for f in flags[4:8]) + (len(flags) > 8 and flags[8] == "true", len(flags) > 9 and flags[9] == "true")
        per_shape[key] = c["fetch_B_per_elem_corrected"] + c["write_B_per_elem"]
    k1 = next((c["fetch_B_per_elem_corrected"] for name, c in kernels.items() if "abs_sum_partials" in name), 4.0)
    total, priced = 0.0, 0
    for rec in sink:
        n_l, e_l, b_l = rec[2], rec[3], rec[4]
        # bytes/element -> which operands moved: 12 (g, d, x read; d written... ) + 4 each for m_in, m_out, x_adv
        shapes = [k for k, v in per_shape.items() if 4 * (4 + sum(k[:4])) == b_l and not k[0]
                  and k[4] == (len(rec) > 5 and bool(rec[5])) and k[5] == (len(rec) > 6 and bool(rec[6]))]
        if shapes:
            total += per_shape[shapes[0]] * n_l * e_l
            priced += 1
    if priced != len(sink):
        return None
    n_, e_ = sink[0][2], sink[0][3]
    ran_k1 = sum(bool(rec[7]) if len(rec) > 7 else _hip.stats["k1_passes"] > 0 for rec in sink) / len(sink)
    return {"bytes_per_launch": int(total / len(sink) + k1 * e_ * n_ * ran_k1),
            "source": "NOT measured in this run: profiles/pmc_update_kernel.json, the committed rocprofv3 --pmc FETCH_SIZE / "
                      "WRITE_SIZE passes over the shipped kernel at N=125, priced per launch"}


def literal_leg(args, _hip):
    """The reference-literal setup, timed beside the headline: what ``main.py`` does with no environment set
    (/root/reference/main.py:15,36) -- batches of 32, NCHW, separate eval-mode BatchNorm, plain nn.Module execution, MIOpen's
    immediate mode (no ``cudnn.benchmark`` find) -- same attack, same surrogate, same synthetic images.  A user who switches
    packages and nothing else gets THIS rate; the headline's arrangement is the two switches TA_FOLD_BN /
    TA_CHANNELS_LAST + MIOpen find.  -> ``config.literal``."""
    import contextlib
    import transferattack_amd as ta
    saved = {k: os.environ.get(k) for k in ("TA_FOLD_BN", "TA_CHANNELS_LAST", "TA_CK_EPILOGUE")}
    saved_benchmark = torch.backends.cudnn.benchmark
    os.environ["TA_FOLD_BN"], os.environ["TA_CHANNELS_LAST"], os.environ["TA_CK_EPILOGUE"] = "0", "0", "0"
    torch.backends.cudnn.benchmark = False
    batch = 32
    try:
        model_name = args.model.split(",") if "," in args.model else args.model
        with contextlib.redirect_stdout(sys.stderr):
            attacker = ta.load_attack_class(args.attack)(model_name=model_name)
        dev = attacker.device
        batches = [tuple(t.to(dev) for t in synthetic_batch(batch, 7000 + 2 * i, args.image_size, args.classes)) for i in range(2)]
        attacker(*batches[0])                                    # warm-up: MIOpen's immediate-mode kernels for these shapes
        torch.cuda.synchronize()
        _hip.profile_sink, note = [], None
        try:
            _hip.timing_begin(args.literal_steps * 64 + 64)
        except _hip.HipExtensionError as exc:
            note = str(exc)[:200]
        t0 = time.perf_counter()
        for i in range(args.literal_steps):
            attacker(*batches[i % 2])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sink, _hip.profile_sink = _hip.profile_sink, None
        dispatch_ms = _hip.timing_end() if note is None else []
        out = {"images_per_s": round(args.literal_steps * batch / dt, 2), "batch": batch, "steps": args.literal_steps,
               "ms_per_step": round(dt / args.literal_steps * 1e3, 2),
               "arrangement": "NCHW, separate BatchNorm, nn.Module execution, MIOpen immediate mode, no environment switches "
                              "(what main.py runs by default)"}
        if sink:
            r = roofline(args, sink, dispatch_ms, note, _hip, byte_source_taken(batches))
            out.update({"update_frac": r["frac"], "update_frac_at_24B_contract": r["frac_at_24B_contract"],
                        "update_mean_us": r["mean_us"], "update_kernel": r["kernel"], "k1_passes": r["k1_passes"],
                        "update_launches": r["launches"]})
        return out
    finally:
        _hip.profile_sink = None
        torch.backends.cudnn.benchmark = saved_benchmark
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def main(argv=None):
    args = parse(argv)
    on_gpu = args.device == "cuda"
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)
    rank, world, local = open_world(args)

    os.environ["TA_FOLD_BN"] = "1" if args.fold_bn else "0"
    os.environ["TA_CHANNELS_LAST"] = "1" if args.channels_last else "0"
    os.environ["TA_CK_EPILOGUE"] = "1" if args.ck_epilogue and args.fold_bn and args.channels_last and on_gpu else "0"
    from transferattack_amd import _ck, _hip
    _hip.load()
    torch.backends.cudnn.benchmark = True                     # MIOpen picks its fastest conv algorithms
    attacker, shard_rank, shard_world, layout = build_attacker(args, world)
    dev = attacker.device

    total = args.steps + args.warmup
    batches = [tuple(t.to(dev) for t in synthetic_batch(args.batch, 1000 * shard_rank + 2 * i, args.image_size, args.classes))
               for i in range(min(total, 4))]

    def step(i):
        x, y = batches[i % len(batches)]
        return attacker(x, y)

    state = {"timing_note": None, "records": {}, "restore": lambda: None}

    def before_timed_steps():
        if not on_gpu:
            return
        _hip.profile_sink = []
        try:
            _hip.timing_begin(args.steps * 64 + 64)     # events on the update kernels' own dispatch packets
        except _hip.HipExtensionError as exc:
            state["timing_note"] = str(exc)[:200]
        _hip.stats["partials_reused"] = _hip.stats["k1_passes"] = 0
        if args.kernel_times:
            state["records"], state["restore"] = instrument_kernels(_hip, lambda: torch.cuda.Event(enable_timing=True))

    t_start = time.perf_counter()
    dt, mine, (waited_s, warmup_s, staged) = timed_region(step, args, world, before_timed_steps, rank,
                                                          independent="no collective" in layout)
    startup_s = time.perf_counter() - t_start - dt
    sink, _hip.profile_sink = _hip.profile_sink, None
    timing_note, kernel_records = state["timing_note"], state["records"]
    dispatch_ms = []
    if on_gpu and timing_note is None:
        try:
            dispatch_ms = _hip.timing_end()
        except _hip.HipExtensionError as exc:
            timing_note = str(exc)[:200]
    state["restore"]()
    dt, per_rank, observed_world, backend = over_ranks(dt, round(args.steps * args.batch / mine, 2), world, dev)

    if rank == 0:
        images = args.steps * args.batch * shard_world
        if on_gpu and not sink:   # attacks that call the two hooks separately: time the fused pair stand-alone
            x0 = batches[0][0]
            g0, m0, d0 = torch.randn_like(x0) * 1e-4, torch.randn_like(x0), torch.zeros_like(x0)
            _hip.profile_sink = sink = []
            if timing_note is None:
                _hip.timing_begin(32)
            for _ in range(20):
                _hip.mi_update(g0, m0, m0, d0, x0, 1.0, 1.6 / 255, 16 / 255)
            torch.cuda.synchronize()
            _hip.profile_sink = None
            dispatch_ms = _hip.timing_end() if timing_note is None else []
        result = {
            "metric": "adversarial images/sec (1000-img set, K=10)",
            "value": round(images / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"fold_bn": os.environ.get("TA_FOLD_BN", "0") == "1",
                       "images_per_step": args.batch,
                       "channels_last": os.environ.get("TA_CHANNELS_LAST", "0") == "1",
                       # execution strategy of a folded ResNet surrogate (backbones/fused.py): same MIOpen convolutions, the
                       # bias / ReLU / residual / threshold passes between them fused, the stem's input gradient on csrc/stem.hip
                       "fused_glue": os.environ.get("TA_FUSED_GLUE", "1") != "0" and os.environ.get("TA_FOLD_BN", "0") == "1",
                       "stem_kernel": os.environ.get("TA_STEM_KERNEL", "1") != "0" and os.environ.get("TA_FOLD_BN", "0") == "1"
                       and os.environ.get("TA_FUSED_GLUE", "1") != "0" and os.environ.get("TA_CHANNELS_LAST", "0") == "1",
                       "workload": "%s on %s (seeded random init), eps=16/255, alpha=1.6/255, K=10, synthetic "
                                   "3x%dx%d, batches of %d, %s"
                                   % ("configs[1]: MI-FGSM" if args.attack == "mifgsm" else args.attack, args.model,
                                      args.image_size, args.image_size, args.batch, layout),
                       # glue passes as epilogues of composable_kernel convolutions (libta_ck.so), site by site where faster
                       "ck_epilogue": {"on": _ck.enabled(), "sites_tuned": _ck.stats["tuned_sites"], "sites_from_plan_file": _ck.stats["sites_from_disk"],
                                       "sites_fused": _ck.stats["sites_on_ck"]},
                       "byte_source": os.environ.get("TA_U8_SOURCE", "1") != "0",
                       # the surrogate's Normalize folded into the two HIP kernels either side of the backbone
                       # (attack.py::_forward_normalize_folded): no x + delta store, no gx = gy / std store
                       "normalize_folded": os.environ.get("TA_FOLD_NORMALIZE", "1") != "0",
                       "attack": args.attack, "surrogate": args.model, "batch": args.batch, "iterations": 10,
                       "parallelism": layout, "gpus_requested": args.gpus, "ranks_observed": observed_world,
                       "collective_backend": backend, "images_per_s_per_rank": per_rank,
                       # before the timed region, on rank 0: its own warm-up steps (MIOpen find on a fresh node) + the other
                       # ranks' warm-up, which starts when rank 0's ends (staged_warmup)
                       "startup": {"rank0_warmup_s": round(warmup_s, 1), "until_timed_region_s": round(startup_s, 1),
                                   "staged": staged}},
            "roofline": roofline(args, sink, dispatch_ms, timing_note, _hip, byte_source_taken(batches)) if sink else None,
        }
        if kernel_records:
            result["config"]["kernels"] = summarise_kernels(kernel_records)
        if on_gpu and args.kernel_sweep and world == 1:
            try:
                result["config"]["update_kernel_sweep"] = kernel_sweep()
            except Exception as exc:  # noqa: BLE001
                result["config"]["update_kernel_sweep"] = {"error": repr(exc)[:200]}
        if on_gpu and args.literal_steps > 0 and world == 1:
            try:
                result["config"]["literal"] = literal_leg(args, _hip)
            except Exception as exc:  # noqa: BLE001  -- a second figure: never at the expense of the line
                result["config"]["literal"] = {"error": repr(exc)[:200]}
        if args.cpu_images > 0 and world == 1:
            try:
                result["cpu_baseline"] = cpu_baseline(args)
            except Exception as exc:  # noqa: BLE001
                result["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port",
                                          "sample": "failed: " + repr(exc)[:200]}
        print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    main()
