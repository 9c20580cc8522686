"""CPU, world_size 2, gloo: the N > 1 host logic.  HIP kernels cannot run here, so the
per-rank pieces use the oracle's arithmetic; what is under test is the product's reduction logic:
global masked mean (sum, count all-reduced BEFORE dividing), one flat-buffer gradient all-reduce, identical
clip / skip decision on every rank, and utterance sharding without a collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import heads as oheads
from oracle import objective as oobj


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch():
    g = torch.Generator().manual_seed(5)
    B, F, D, N = 4, 12, 10, 7
    feats = torch.randn(B, F, D, generator=g)
    lin = torch.rand(B, F, N, generator=g) + 0.1
    tar = torch.rand(B, F, N, generator=g) + 0.1
    lens = torch.tensor([12, 5, 9, 1])                # ragged: a mean of per-rank means would be wrong
    W = torch.randn(N, D, generator=g) * 0.3
    b = torch.randn(N, generator=g) * 0.1
    return feats, lin, tar, lens, W, b


def _loss_terms(feats, lin, tar, lens, W, b):
    pred, _ = oheads.linear_residual(feats, lin, W, b)
    masks = (torch.arange(feats.shape[1])[None] < lens[:, None]).long()
    return oobj.l1_sums((pred + 1e-10).log(), tar, masks)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from speech_enhancement_by_s3prl_amd import dist as sdist
    feats, lin, tar, lens, W, b = _make_batch()
    idx = sdist.shard_indices(feats.shape[0])
    W = W.clone().requires_grad_(True)
    b = b.clone().requires_grad_(True)
    s, n = _loss_terms(feats[idx], lin[idx], tar[idx], lens[idx], W, b)
    sums = sdist.all_reduce_sums(torch.stack([s.detach().double(), n.double()]))
    loss_global = (sums[0] / sums[1]).item()
    (s / sums[1].float()).backward()                  # local gradient already scaled by 1 / GLOBAL count
    # replicas that start different are made equal (rank 0 wins), buffers included
    m = torch.nn.BatchNorm1d(3)
    with torch.no_grad():
        m.weight.fill_(float(rank + 1))
        m.running_mean.fill_(float(10 * rank + 5))
    sdist.broadcast_parameters(m)
    assert torch.equal(m.weight, torch.ones(3)) and torch.equal(m.running_mean, torch.full((3,), 5.0))
    unused = torch.ones(3, requires_grad=True)       # a parameter no loss term touches: grad stays None on every rank
    red = sdist.FlatGradAllReducer([W, b, unused])
    flat = red.reduce()[:W.numel() + b.numel()].clone()
    assert red.active == [True, True, False] and unused.grad is None      # BertAdam's `if p.grad is None: continue` still applies
    gn = torch.nn.utils.clip_grad_norm_([W, b], 1.0)
    # numpy arrays, not tensors: a tensor travels through a multiprocessing queue as a shared-memory file descriptor that the RECEIVER fetches from the
    # sender over a socket -- if this process has exited by then the parent sees ConnectionResetError (seen once in round 4 on a loaded box)
    q.put((rank, idx, loss_global, flat.numpy().copy(), float(gn), W.grad.numpy().copy(), b.grad.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_matches_single_process():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    res = [(r[0], r[1], r[2], torch.from_numpy(r[3]), r[4], torch.from_numpy(r[5]), torch.from_numpy(r[6])) for r in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process truth
    feats, lin, tar, lens, W, b = _make_batch()
    W = W.clone().requires_grad_(True)
    b = b.clone().requires_grad_(True)
    s, n = _loss_terms(feats, lin, tar, lens, W, b)
    loss = s / n
    loss.backward()
    ref_flat = torch.cat([W.grad.flatten(), b.grad.flatten()])
    assert res[0][1] == [0, 2] and res[1][1] == [1, 3]                     # utterance i -> rank i % world
    for r in res:
        assert abs(r[2] - loss.item()) < 1e-6                               # global masked mean, not mean of means
        assert torch.allclose(r[3], ref_flat, rtol=1e-5, atol=1e-7)         # exact single-process gradient
    assert res[0][4] == pytest.approx(res[1][4], rel=1e-6)                  # identical clip decision on every rank
    assert torch.equal(res[0][5], res[1][5]) and torch.equal(res[0][6], res[1][6])
    # and it differs from the naive mean of per-rank means (the bug this design avoids)
    naive = 0.5 * sum((_loss_terms(feats[i], lin[i], tar[i], lens[i], W.detach(), b.detach())[0] /
                       _loss_terms(feats[i], lin[i], tar[i], lens[i], W.detach(), b.detach())[1]).item()
                      for i in ([0, 2], [1, 3]))
    assert abs(naive - loss.item()) > 1e-4


def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` outside torchrun must start 2 ranks (fresh child `python -m torch.distributed.run`, nothing
    exec'd after a GPU call) and report the world size the ranks OBSERVED; the rank plumbing is exercised over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--launch-check', '--backend', 'gloo'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                           # ONE JSON line on stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['world_observed'] == 2


def _run_bench(argv, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=timeout)


@pytest.mark.parametrize('workload,world', [('finetune', 2), ('enhance', 2), ('finetune', 4)])
def test_bench_control_flow_two_ranks_stubbed(workload, world):
    """The WHOLE control flow of bench.py at N = 2 (and, for the all-reducing step, N = 4) over gloo with the step stubbed (`--stub-step`: same collective sequence per step --
    (sum, count) all-reduce, per-layer async bucket all-reduces, tail reduce + wait -- through the product's own dist.py classes): warm-up,
    timed loop, max-over-ranks, the roofline leg on EVERY rank, teardown, then rank 0's side legs with no process group left.  Round 2's
    2-rank fine-tune rehearsal deadlocked because rank 0 alone re-ran all-reducing steps; this is the test that would have caught it
    (a hang here fails by timeout)."""
    import json
    out = _run_bench(['--gpus', str(world), '--workload', workload, '--stub-step', '--backend', 'gloo', '--steps', '3', '--warmup', '1'], timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == world and rec['steps'] == 3 and 'stub' in rec and rec['value'] > 0
    assert rec['config']['global_batch'] == world * rec['config']['batch_per_gpu']


def test_bench_phase_b_is_collective_free():
    """static check of the invariant the fix rests on: in bench.main() nothing between Ranks() and ranks.teardown() is conditional on the
    rank, and everything rank-0-only sits after teardown()."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'bench.py')).read()
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'main')
    start = next(i for i, n in enumerate(main.body) if isinstance(n, ast.Assign) and 'Ranks(args)' in ast.get_source_segment(src, n))
    end = next(i for i, n in enumerate(main.body) if isinstance(n, ast.Expr) and 'ranks.teardown()' in ast.get_source_segment(src, n))
    assert start < end
    for node in main.body[start:end]:
        for sub in ast.walk(node):
            if isinstance(sub, (ast.If, ast.IfExp, ast.While)):
                cond = ast.get_source_segment(src, sub.test)
                assert 'rank' not in cond.replace('ranks.', '').replace('do_roofline', ''), f'rank-conditional code inside the collective phase: {cond}'
    tail = '\n'.join(ast.get_source_segment(src, n) for n in main.body[end + 1:])
    assert 'if rank != 0' in tail and 'host_fed_leg' in tail and 'cpu_baseline' in tail
