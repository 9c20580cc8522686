"""GPU parity of the LinearResidual training step pieces (runner.py:453-471): head backward through the HIP
kernels vs the reference's own autograd (golden chain gradient) and the oracle at full size."""
import pytest
import torch

from oracle import heads as oheads
from oracle import objective as oobj

pytestmark = pytest.mark.gpu


def T(x, dev=None):
    t = torch.from_numpy(x)
    return t.to(dev) if dev is not None else t


def test_chain_gradient_golden(gpu, golden):
    """loss = L1(log(LinearResidual(feats, linears) + 1e-10), tar); d loss / d (W, b) from the reference's autograd."""
    from speech_enhancement_by_s3prl_amd.decode import get_length_masks
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    from speech_enhancement_by_s3prl_amd.objective import L1
    m = LinearResidual(input_size=120, output_size=201, cmvn=True).to(gpu)
    m.load_state_dict({'linear.weight': T(golden['c1_weight']), 'linear.bias': T(golden['c1_bias'])})
    pred, _ = m(features=T(golden['c1_feats'], gpu), linears=T(golden['c1_linears'], gpu))
    logp = (pred + 1e-10).log()
    masks = get_length_masks(T(golden['d1_lengths'], gpu))
    loss, _ = L1()(log_predicted=logp, linear_tar=T(golden['e1_linear_tar'], gpu), stft_length_masks=masks)
    loss.backward()
    assert abs(loss.item() - float(golden['chain_loss'])) < 1e-5 * abs(float(golden['chain_loss']))
    gw, gb = T(golden['chain_gw']), T(golden['chain_gb'])
    assert (m.linear.weight.grad.cpu() - gw).abs().max().item() < 1e-4 * gw.abs().max().item()
    assert (m.linear.bias.grad.cpu() - gb).abs().max().item() < 1e-4 * gb.abs().max().item()


@pytest.mark.parametrize('B,F,D,N,act', [(2, 1001, 120, 201, 'Sigmoid'), (1, 300, 768, 201, 'ReLU'), (3, 50, 40, 33, 'Identity')])
def test_head_backward_vs_autograd(gpu, B, F, D, N, act):
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    torch.manual_seed(F)
    feats, lin, g = torch.randn(B, F, D), torch.rand(B, F, N) + 0.1, torch.randn(B, F, N)
    m = LinearResidual(input_size=D, output_size=N, activation=act, cmvn=True)
    w = m.linear.weight.detach().clone().double().requires_grad_(True)
    b = m.linear.bias.detach().clone().double().requires_grad_(True)
    rp, _ = oheads.linear_residual(feats.double(), lin.double(), w, b, activation=act)
    (rp * g.double()).sum().backward()
    m = m.to(gpu)
    p, _ = m(features=feats.to(gpu), linears=lin.to(gpu))
    (p * g.to(gpu)).sum().backward()
    assert (m.linear.weight.grad.cpu().double() - w.grad).abs().max().item() < 1e-4 * w.grad.abs().max().item()
    assert (m.linear.bias.grad.cpu().double() - b.grad).abs().max().item() < 1e-4 * b.grad.abs().max().item()


def _opt_fixture(device, seed=0):
    from speech_enhancement_by_s3prl_amd.solver import get_optimizer
    g = torch.Generator().manual_seed(seed)
    shapes = {'a.weight': (201, 120), 'a.bias': (201,), 'enc.LayerNorm.weight': (768,), 'enc.LayerNorm.bias': (768,),
              'big.weight': (3072, 768), 'odd.weight': (37, 5), 'tiny.bias': (3,)}
    named = [(n, torch.nn.Parameter((torch.randn(*s, generator=g) * 0.1).to(device))) for n, s in shapes.items()]
    opt = get_optimizer(named, lr=1e-3, warmup_proportion=0.1, training_steps=20)
    grads = [[torch.randn(*s, generator=g) * sc for s in shapes.values()] for sc in (0.001, 0.2, 3.0, 0.05)]   # per-tensor clip on and off
    return named, opt, grads


def test_fused_bertadam_matches_oracle(gpu):
    """two-launch BertAdam (se_multi_sumsq_f32 + se_bertadam_step_f32) vs oracle/optim.py in float64: parameters and both
    moments after 4 steps (fp32 bound 1e-5); the per-tensor torch path on the CPU agrees to 2e-4 only -- its fp32 norm of a
    2.4 M element tensor is itself 4e-5 off."""
    from oracle import optim as oopt
    named_g, opt_g, grads = _opt_fixture(gpu)
    named_c, opt_c, _ = _opt_fixture(torch.device('cpu'))
    ref = {n: [p.detach().cpu().double(), torch.zeros(p.shape, dtype=torch.float64), torch.zeros(p.shape, dtype=torch.float64)] for n, p in named_g}
    for k, step_grads in enumerate(grads):
        for (n, pg), (_, pc), g in zip(named_g, named_c, step_grads):
            pg.grad, pc.grad = g.to(gpu), g.clone()
            wd = 0.0 if ('bias' in n or 'LayerNorm' in n) else 0.01
            ref[n] = list(oopt.bert_adam_step(ref[n][0], g.double(), ref[n][1], ref[n][2], k, 1e-3, 0.1, 20, wd))
        v0 = named_g[0][1]._version
        opt_g.step()
        opt_c.step()
        assert named_g[0][1]._version > v0           # engines key their bf16 copies on the version counter
    for (n, pg), (_, pc) in zip(named_g, named_c):
        sg = opt_g.state[pg]
        assert sg['step'] == 4
        for got, want, what in ((pg.detach(), ref[n][0], 'p'), (sg['next_m'], ref[n][1], 'm'), (sg['next_v'], ref[n][2], 'v')):
            err = (got.double().cpu() - want).abs().max().item()
            assert err < 1e-5 * want.abs().max().item() + 1e-12, (n, what, err)
        assert torch.allclose(pg.detach().cpu(), pc.detach(), rtol=2e-4, atol=1e-6), n


def test_fused_train_step_global_clip(gpu):
    """DataParallelTrainStep's device path (global clip folded into the update) vs clip_grad_norm_ + BertAdam on the CPU."""
    from speech_enhancement_by_s3prl_amd.dist import DataParallelTrainStep
    from speech_enhancement_by_s3prl_amd.objective import L1
    named_g, opt_g, grads = _opt_fixture(gpu, seed=1)
    named_c, opt_c, _ = _opt_fixture(torch.device('cpu'), seed=1)
    model = torch.nn.Module()
    for i, (_, p) in enumerate(named_g):
        model.register_parameter(f'p{i}', p)
    dp = DataParallelTrainStep(model, L1(), opt_g, grad_clip=1.0)
    for step_grads in grads:
        # a loss whose gradient wrt each parameter is exactly the prescribed tensor
        loss = sum((p * g.to(gpu)).sum() for (_, p), g in zip(named_g, step_grads))
        gn, skipped = dp.step(loss)
        for (_, pc), g in zip(named_c, step_grads):
            pc.grad = g.clone()
        ref_gn = torch.nn.utils.clip_grad_norm_([p for _, p in named_c], 1.0)
        opt_c.step()
        assert not skipped and abs(gn - float(ref_gn)) < 2e-4 * float(ref_gn)
    for (n, pg), (_, pc) in zip(named_g, named_c):
        assert torch.allclose(pg.detach().cpu(), pc.detach(), rtol=3e-4, atol=1e-6), n
    # NaN gradient: the step is skipped, parameters untouched
    before = [p.detach().clone() for _, p in named_g]
    loss = sum((p * float('nan')).sum() for _, p in named_g)
    gn, skipped = dp.step(loss)
    assert skipped and all(torch.equal(b, p.detach()) for b, (_, p) in zip(before, named_g))


def test_multi_copy_many_tensors(gpu):
    """se_multi_copy_f32: more tensors than one kernel-argument table holds (96), ragged sizes, unaligned sources and
    destinations, sizes spanning several 16 Ki-element chunks -- bit-exact copies, untouched neighbours."""
    import ctypes
    from speech_enhancement_by_s3prl_amd import _lib as L
    lib = L.load()
    torch.manual_seed(5)
    sizes = [1, 2, 3, 5, 16384, 16385, 40000, 7] + [int(x) for x in torch.randint(1, 3000, (200,))]
    src_flat = torch.randn(sum(sizes) + len(sizes), device=gpu)
    dst_flat = torch.full((sum(sizes) + 2 * len(sizes),), -7.0, device=gpu)
    srcs, dsts, so, do = [], [], 0, 1
    for n in sizes:                                  # sources packed back to back (+1: misaligned), destinations with a guard element between
        srcs.append(src_flat[so:so + n])
        dsts.append(dst_flat[do:do + n])
        so += n + 1
        do += n + 2
    n = len(sizes)
    L.check(lib.se_multi_copy_f32((ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts]), (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs]),
                                  (ctypes.c_uint64 * n)(*sizes), n, L.stream()), 'se_multi_copy_f32')
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        assert torch.equal(s, d)
    guard = torch.ones_like(dst_flat, dtype=torch.bool)
    do = 1
    for n_ in sizes:
        guard[do:do + n_] = False
        do += n_ + 2
    assert (dst_flat[guard] == -7.0).all()
