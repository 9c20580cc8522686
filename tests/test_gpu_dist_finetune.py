"""Two ranks on ONE MI355X (gloo over device tensors; RCCL needs a device per rank) running the real HIP training path:
a data-parallel Mockingjay fine-tune step on two half batches must reproduce the single-process step on the whole ragged
batch -- global masked mean (the (sum, count) pair is all-reduced before dividing), the flat-buffer gradient all-reduce
issued bucket by bucket from the encoder backward's per-layer callback (dist.BucketedGradSink) plus one for the spec head,
identical gradient norm / skip decision, identical fused BertAdam update."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(device):
    from speech_enhancement_by_s3prl_amd import pipeline
    from speech_enhancement_by_s3prl_amd.solver import get_optimizer
    cfg = pipeline.make_config(layers=2)
    ckpt = pipeline.synthetic_checkpoint(cfg, seed=21)
    model = pipeline.build_mockingjay(ckpt, device)
    model.eval()                       # dropout off: the two runs must see the same function (gradients stay enabled)
    opt = get_optimizer(list(model.named_parameters()), lr=1e-4, warmup_proportion=0.07, training_steps=100)
    g = torch.Generator().manual_seed(3)
    B, T = 4, 96
    feats = torch.randn(B, T, 80, generator=g)
    lens = torch.tensor([96, 40, 77, 13])
    for b in range(B):
        feats[b, lens[b]:] = 0.0
    tar = torch.rand(B, T, 201, generator=g) + 0.05
    return model, opt, feats, tar, lens


def _step(model, opt, feats, tar, lens, device):
    from speech_enhancement_by_s3prl_amd.dist import DataParallelTrainStep
    from speech_enhancement_by_s3prl_amd.objective import L1
    crit = L1()
    dp = DataParallelTrainStep(model, crit, opt, grad_clip=1.0)
    pred, res = model(features=feats.to(device))
    loss, _ = crit(log_predicted=res['log_predicted'], linear_tar=tar.to(device), stft_lengths=lens.to(device))
    gn, skipped = dp.step(loss)
    # the encoder trunk's gradients went through the bucketed sink: written in place (no .grad clone), one bucket per layer + the input stage
    assert dp.sink is not None and dp.sink.buckets == 3, dp.sink and dp.sink.buckets
    trunk = sum(1 for _ in model.mockingjay.model.parameters())
    assert len(dp.sink.done) == trunk, (len(dp.sink.done), trunk)
    assert all(p.grad is None for p in model.mockingjay.model.parameters())      # fused step: gradients are read from the flat buffer
    if dist.is_initialized():
        assert dp.sink.collectives == 3, dp.sink.collectives
    # numpy (pickled by value): torch tensors travel through an mp.Queue by file descriptor, which dies with the worker
    return float(loss.detach()), gn, skipped, dp.reducer.flat.detach().cpu().numpy().copy(), [p.detach().cpu().numpy().copy() for p in model.parameters()]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    model, opt, feats, tar, lens = _setup(dev)
    idx = list(range(rank, feats.shape[0], world))          # utterance i -> rank i % world (dist.shard_indices)
    out = _step(model, opt, feats[idx], tar[idx], lens[idx], dev)
    q.put((rank,) + out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_finetune_step_matches_single_process(gpu):
    model, opt, feats, tar, lens = _setup(gpu)
    loss1, gn1, sk1, flat1, params1 = _step(model, opt, feats, tar, lens, gpu)
    assert not sk1 and gn1 > 0
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, loss, gn, sk, flat, params in got:
        assert not sk
        assert abs(loss - loss1) < 1e-5 * abs(loss1)                       # the criterion returns the GLOBAL masked mean
        assert abs(gn - gn1) < 2e-3 * gn1
        rel = float(np.linalg.norm(flat - flat1) / np.linalg.norm(flat1))
        assert rel < 2e-3, rel              # same bf16 kernels, different fp32 summation splits / atomic orders
    # both ranks applied the identical update
    for a, b in zip(got[0][5], got[1][5]):
        assert np.array_equal(a, b)
    # and it is the single-process update (BertAdam's first step is ~ lr * sign(g): compare where |g| is not ~0)
    moved = sum(int((np.abs(a - c) > 0).sum()) for a, c in zip(got[0][5], params1))
    total = sum(a.size for a in params1)
    assert moved < 0.02 * total, (moved, total)
