"""Command-line driver with the reference's flags (main.py:10-26) on the MI355X engine.

    python main.py --input_dir ./data --output_dir adv_data/mifgsm/resnet50 --attack mifgsm --model=resnet50
    python main.py --input_dir ./data --output_dir adv_data/mifgsm/resnet50 --eval
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py --attack dts ...   (8 GPUs)

Same behaviour as the reference: the hyper-parameter flags --epoch/--eps/--alpha/--momentum/--random_start are
parsed but, as in the reference (main.py:41), not forwarded -- every attack runs with its class defaults.
Added flags: --seed (per-batch seeding so results do not depend on the GPU count), --resume (skip finished batches).
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
    parser.add_argument('--resume', action='store_true',
                        help='skip the batches whose output files all exist already (an interrupted run picks up where it '
                             'stopped; the per-batch seeding makes the remaining batches come out the same)')
    return parser.parse_args()


def main():
    args = get_parser()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", args.GPU_ID)
    rank, world = tadist.init()
    os.makedirs(args.output_dir, exist_ok=True)

    dataset = AdvDataset(input_dir=args.input_dir, output_dir=args.output_dir, targeted=args.targeted, eval=args.eval)
    num_batches = (len(dataset) + args.batchsize - 1) // args.batchsize

    decoders = ThreadPoolExecutor(max_workers=args.io_threads)        # PNG decode (PIL releases the GIL)
    device = default_device()
    copy_stream = torch.cuda.Stream(device) if device.type == "cuda" else None

    def batch(idx):
        """decode one reference batch on the host threads and start its upload: page-locked staging buffer, asynchronous
        copy on a side stream (runs under the previous batch's kernels); returns the event the consumer waits for"""
        lo, hi = idx * args.batchsize, min((idx + 1) * args.batchsize, len(dataset))
        items = list(decoders.map(dataset.__getitem__, range(lo, hi)))
        images = torch.stack([it[0] for it in items])
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
        io = ThreadPoolExecutor(max_workers=2)
        pending_write, next_batch = None, io.submit(batch, mine[0]) if mine else None
        for pos, batch_idx in enumerate(tqdm.tqdm(mine, disable=rank != 0)):
            images, labels, filenames = consume(next_batch.result())
            if pos + 1 < len(mine):
                next_batch = io.submit(batch, mine[pos + 1])
            tadist.seed_batch(args.seed, batch_idx)
            perturbations = attacker(images, labels)
            if pending_write is not None:
                pending_write.result()
            if writer:
                pending_write = io.submit(save_images, args.output_dir, images, filenames, perturbations)
        if pending_write is not None:
            pending_write.result()
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
