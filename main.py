"""Command-line driver with the reference's flags (main.py:10-26) on the MI355X engine.

    python main.py --input_dir ./data --output_dir adv_data/mifgsm/resnet50 --attack mifgsm --model=resnet50
    python main.py --input_dir ./data --output_dir adv_data/mifgsm/resnet50 --eval
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py --attack dts ...   (8 GPUs)

Same behaviour as the reference: the hyper-parameter flags --epoch/--eps/--alpha/--momentum/--random_start are
parsed but, as in the reference (main.py:41), not forwarded -- every attack runs with its class defaults.
Added flags: --seed (per-batch seeding so results do not depend on the GPU count), --resume (skip finished batches),
--coalesce K (attacks that treat the images of a batch independently -- transferattack_amd.BATCH_INDEPENDENT -- run K
reference batches per device batch: the per-image L1 normalisation and the sign step make the result independent of how the
loss is averaged over the batch, and a 128-image launch uses the GPU better than four 32-image ones -- at 4x the activation
memory, and MIOpen may pick other algorithms for the larger batch, so "same result" means same arithmetic per image, not a
bit guarantee on the device; DIM / Admix / ... keep the reference's batches, and attacks that draw noise on the device
(VMI / VNI neighbours, random starts) default to K = 1 so that an image's draws do not depend on the grouping), --profile (seconds per pipeline stage at the end).
Input pipeline: threaded PNG decode -> page-locked staging buffer -> asynchronous upload on a side stream while the previous
batch runs; output: GPU quantiser -> uint8 download -> threaded PNG encode, also overlapped.  With several processes the
dataset is sharded by whole batches (transferattack_amd.dist.shard_batches); for ``--attack ens`` every group of
len(models) ranks holds one surrogate each and exchanges logits / input-gradients over RCCL.
"""
import argparse
import os
from concurrent.futures import ThreadPoolExecutor

import torch
import tqdm

import transferattack_amd as transferattack
from transferattack_amd import dist as tadist
from transferattack_amd.utils import (AdvDataset, cnn_model_paper, load_pretrained_model, save_images,
                                      vit_model_paper, wrap_model, default_device)


def get_parser():
    parser = argparse.ArgumentParser(description='Generating transferable adversaria examples')
    parser.add_argument('-e', '--eval', action='store_true', help='attack/evluation')
    parser.add_argument('--attack', default='mifgsm', type=str, help='the attack algorithm',
                        choices=transferattack.attack_zoo.keys())
    parser.add_argument('--epoch', default=10, type=int, help='the iterations for updating the adversarial patch')
    parser.add_argument('--batchsize', default=32, type=int, help='the bacth size')
    parser.add_argument('--eps', default=16 / 255, type=float, help='the stepsize to update the perturbation')
    parser.add_argument('--alpha', default=1.6 / 255, type=float, help='the stepsize to update the perturbation')
    parser.add_argument('--momentum', default=0., type=float, help='the decay factor for momentum based attack')
    parser.add_argument('--model', default='resnet50', type=str, help='the source surrogate model')
    parser.add_argument('--ensemble', action='store_true', help='enable ensemble attack')
    parser.add_argument('--random_start', default=False, type=bool, help='set random start')
    parser.add_argument('--input_dir', default='./data', type=str,
                        help='the path for custom benign images, default: untargeted attack data')
    parser.add_argument('--output_dir', default='./results', type=str, help='the path to store the adversarial patches')
    parser.add_argument('--targeted', action='store_true', help='targeted attack')
    parser.add_argument('--GPU_ID', default='0', type=str)
    parser.add_argument('--seed', default=0, type=int, help='base seed of the per-batch host RNG (DIM / Admix draws)')
    parser.add_argument('--io_threads', default=4, type=int, help='host threads decoding / encoding PNGs')
    parser.add_argument('--coalesce', default=0, type=int,
                        help='reference batches per device batch for batch-independent attacks (0 = 4 for those, 1 otherwise)')
    parser.add_argument('--profile', action='store_true', help='print the busy seconds of every pipeline stage at the end')
    parser.add_argument('--resume', action='store_true',
                        help='skip the batches whose output files all exist already (an interrupted run picks up where it '
                             'stopped; the per-batch seeding makes the remaining batches come out the same)')
    return parser.parse_args()


def main():
    args = get_parser()
    # convolution algorithms: MIOpen's immediate mode by default (a heuristic pick per layer, only the chosen solver is built --
    # a fresh box starts in seconds); TA_CONV_TUNE=1 = torch.backends.cudnn.benchmark: every applicable solver is built and
    # timed once per (layer, batch) -- minutes on a fresh box, faster steady state (tools/cold_start.py has both)
    if os.environ.get("TA_CONV_TUNE", "0") == "1":
        torch.backends.cudnn.benchmark = True
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", args.GPU_ID)
    rank, world = tadist.init()
    os.makedirs(args.output_dir, exist_ok=True)

    dataset = AdvDataset(input_dir=args.input_dir, output_dir=args.output_dir, targeted=args.targeted, eval=args.eval)
    num_batches = (len(dataset) + args.batchsize - 1) // args.batchsize

    decoders = ThreadPoolExecutor(max_workers=args.io_threads)        # PNG decode (PIL releases the GIL)
    import transferattack_amd.utils as ta_utils
    ta_utils.IO_THREADS = args.io_threads                             # ... and as many for the encoder
    device = default_device()
    copy_stream = torch.cuda.Stream(device) if device.type == "cuda" else None

    import threading
    import time
    stage_seconds, stage_lock = {}, threading.Lock()

    def spent(stage, t0):
        with stage_lock:
            stage_seconds[stage] = stage_seconds.get(stage, 0.0) + time.perf_counter() - t0

    def batch(idx):
        """decode one device batch (``idx``: a reference-batch index or a list of them) on the host threads and start its
        upload: page-locked staging buffer, asynchronous copy on a side stream (runs under the previous batch's kernels);
        returns the event the consumer waits for"""
        t0 = time.perf_counter()
        wanted = []
        for i in ([idx] if isinstance(idx, int) else idx):
            wanted.extend(range(i * args.batchsize, min((i + 1) * args.batchsize, len(dataset))))
        items = list(decoders.map(dataset.__getitem__, wanted))
        images = torch.stack([it[0] for it in items])
        spent("decode (PNG -> fp32 NCHW, %d host threads)" % args.io_threads, t0)
        t0 = time.perf_counter()
        if args.targeted:
            labels = [torch.tensor([it[1][0] for it in items]), torch.tensor([it[1][1] for it in items])]
        else:
            labels = torch.tensor([it[1] for it in items])
        ready = None
        if copy_stream is not None:
            with torch.cuda.device(device), torch.cuda.stream(copy_stream):        # the device is thread-local state
                staged = images.pin_memory()
                images = staged.to(device, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(copy_stream)
                images._ta_staging = staged                                       # keep the pinned buffer until consumed
        spent("stage + enqueue upload (pinned copy, async H2D)", t0)
        return images, labels, [it[2] for it in items], ready

    def consume(loaded):
        """order the consumer's stream after the upload of ``loaded`` (a result of ``batch``) and tell the caching
        allocator that this stream uses the tensor too (it was allocated on the copy stream)"""
        images, labels, filenames, ready = loaded
        if ready is not None:
            current = torch.cuda.current_stream(device)
            current.wait_event(ready)
            images.record_stream(current)
        return images, labels, filenames

    if not args.eval:
        if args.ensemble or len(args.model.split(',')) > 1:
            args.model = args.model.split(',')
        shard_rank, shard_world = rank, world
        if isinstance(args.model, list) and world >= len(args.model) and world % len(args.model) == 0 and world > 1:
            # one surrogate per rank of a model group (RCCL logit / gradient all-reduce inside ShardedEnsemble.forward for
            # every attack class; per-member broadcast for the attacks that address single members), image shards
            # across groups
            attacker, member, shard_rank, shard_world = tadist.sharded_attack(
                transferattack.load_attack_class(args.attack), args.attack, args.model, world, targeted=args.targeted)
            writer = member == 0
        else:
            attacker = transferattack.load_attack_class(args.attack)(model_name=args.model, targeted=args.targeted)
            writer = True
        # three-stage software pipeline: batch i+1 is decoded and batch i-1 is quantised / PNG-encoded on host
        # threads while batch i runs on the GPU (the reference decodes in DataLoader workers and writes inline)
        mine = tadist.shard_batches(num_batches, shard_rank, shard_world)
        if args.resume:
            def done(idx):
                lo, hi = idx * args.batchsize, min((idx + 1) * args.batchsize, len(dataset))
                return all(os.path.isfile(os.path.join(args.output_dir, dataset.filenames[i])) for i in range(lo, hi))
            mine = [idx for idx in mine if not done(idx)]
            if world > 1:                      # every rank decides from the same directory listing before anyone writes
                torch.distributed.barrier()
        # device batches: K reference batches at a time where the attack does not couple the images of a batch
        # default: 4 for the batch-independent attacks that draw nothing on the device; 1 for those that do (VMI / VNI
        # neighbours, --random_start: the in-kernel Philox counter runs over the flat element index of the DEVICE batch, so
        # an image's draws would depend on how batches were grouped -- i.e. on the GPU count and on what --resume skipped)
        draws_on_device = args.attack in transferattack.DEVICE_NOISE or getattr(attacker, "random_start", False)
        k = args.coalesce if args.coalesce > 0 else (4 if args.attack in transferattack.BATCH_INDEPENDENT
                                                     and not isinstance(args.model, list) and not draws_on_device else 1)
        if args.attack not in transferattack.BATCH_INDEPENDENT:
            k = 1
        # groups of 4 / 2 / 1 FULL reference batches (a short last batch stays alone): the batch-mean loss then carries
        # 1/(kB) instead of 1/B, an exact power-of-two factor that the per-image normalisation removes without a rounding
        # difference -- the coalesced run computes the reference batches' arithmetic, not just its value
        full = [idx for idx in mine if (idx + 1) * args.batchsize <= len(dataset)]
        groups, at = [], 0
        while at < len(full):
            take = 1
            while take * 2 <= k and at + take * 2 <= len(full):
                take *= 2
            groups.append(full[at:at + take])
            at += take
        groups += [[idx] for idx in mine if (idx + 1) * args.batchsize > len(dataset)]

        def write(images, filenames, perturbations):
            t0 = time.perf_counter()
            save_images(args.output_dir, images, filenames, perturbations)
            spent("quantise on the GPU + download + PNG encode", t0)

        io = ThreadPoolExecutor(max_workers=2)
        attack_events = []
        wall0 = time.perf_counter()
        pending_write, next_batch = None, io.submit(batch, groups[0]) if groups else None
        for pos, group in enumerate(tqdm.tqdm(groups, disable=rank != 0)):
            t0 = time.perf_counter()
            images, labels, filenames = consume(next_batch.result())
            spent("main thread waits for decode / upload", t0)
            if pos + 1 < len(groups):
                next_batch = io.submit(batch, groups[pos + 1])
            tadist.seed_batch(args.seed, group[0])
            if args.profile and device.type == "cuda":
                attack_events.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                attack_events[-1][0].record()
            perturbations = attacker(images, labels)
            if args.profile and device.type == "cuda":
                attack_events[-1][1].record()
            t0 = time.perf_counter()
            if pending_write is not None:
                pending_write.result()
            spent("main thread waits for the previous batch's write", t0)
            if writer:
                pending_write = io.submit(write, images, filenames, perturbations)
        if pending_write is not None:
            pending_write.result()
        if args.profile and rank == 0:
            if device.type == "cuda":
                torch.cuda.synchronize()
                stage_seconds["attack on the GPU (K iterations; HIP events)"] = sum(a.elapsed_time(b) for a, b in attack_events) / 1e3
            wall = time.perf_counter() - wall0
            images_done = sum(min((i + 1) * args.batchsize, len(dataset)) - i * args.batchsize for g in groups for i in g)
            import json
            print(json.dumps({"end_to_end_images_per_s": round(images_done / max(wall, 1e-9), 2), "images": images_done,
                              "wall_s": round(wall, 3), "reference_batches_per_device_batch": k,
                              "device_batches": [len(g) for g in groups],
                              "stage_busy_seconds": {k_: round(v, 3) for k_, v in sorted(stage_seconds.items())}}))
    elif rank == 0:
        if not os.environ.get("TA_WEIGHTS_DIR") and os.environ.get("TA_ALLOW_RANDOM_INIT", "0") != "1":
            raise SystemExit("--eval needs the victims' pretrained weights (TA_WEIGHTS_DIR=<dir with <name>.pth>): the attack "
                             "success rate against seeded random-init victims says nothing about the reference's numbers "
                             "(TA_ALLOW_RANDOM_INIT=1 runs it anyway, e.g. for plumbing tests)")
        res = '|'
        for model_name, model in load_pretrained_model(cnn_model_paper, vit_model_paper):
            model = wrap_model(model.eval().to(default_device()))
            for p in model.parameters():
                p.requires_grad = False
            asr = evaluate(model, (consume(batch(i)) for i in range(num_batches)), args.targeted)
            print(f'{model_name}: {asr:.1f}')
            res += f' {asr:.1f} |'
        print(res)
        with open('results_eval.txt', 'a') as f:
            f.write(args.output_dir + res + '\n')


def evaluate(model, batches, is_targeted):
    """ASR on the saved uint8 images (main.py:80-94): untargeted (1 - correct/total)*100, targeted hit rate."""
    correct, total = 0, 0
    dev = next(model.parameters()).device
    with torch.no_grad():
        for images, labels, _ in batches:
            if is_targeted:
                labels = labels[1]
            pred = model(images.to(dev))                      # already there when the pipeline uploaded it
            correct += int((labels.numpy() == pred.argmax(dim=1).cpu().numpy()).sum())
            total += labels.shape[0]
    return (correct / total) * 100 if is_targeted else (1 - correct / total) * 100


if __name__ == '__main__':
    main()
