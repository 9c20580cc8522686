#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X speech-enhancement hot path.

Metric (BASELINE.json): enhanced 10 s utterances / second.  Workload at every N = configs[1]:
"TERA-base upstream + 2-layer mask head on libri-test-clean-10s, batch 32, bf16, 1 x MI355X", i.e. one
evaluate()-style pass (runner.py:556-575) per step over a batch of 32 synthetic 10 s / 16 kHz utterances
already resident in HBM:  STFT of the noisy + clean channels -> mel/log/delta/CMVN features -> 6-layer
768/12/3072 encoder (the only architecture config the reference ships, config/pretrain_sample.yaml:1-22)
-> TransformerSpecPredictionHead + exp/ReLU -> iSTFT with the noisy phase -> level normalisation to the
clean wav -> masked log-L1 loss.  Weights: seeded random at the real sizes (no checkpoints exist offline).

Multi-GPU: utterances shard data-parallel with NO data-path collective (inference); every rank runs
the same per-GPU batch ("weak" scaling); value = all utterances / max-over-ranks time.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...            (no torchrun environment: starts the N ranks itself, see launch_ranks())
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Prints ONE JSON line on rank 0 (`n_gpus` = the world size the ranks OBSERVED through the collective, checked against --gpus).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16


def cpu_baseline(batch, layers, seconds_budget=25.0):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores, bounded sample."""
    import oracle
    from oracle import decode as odec, encoder as oenc, heads as oheads, objective as oobj, preprocessor as opre
    from speech_enhancement_by_s3prl_amd import pipeline, synth
    # threads actually used: this process's CPU share (affinity), capped at 16 -- the GPU box gives one GPU job a
    # 16-core share of a 256-thread host; asking torch for all 256 oversubscribes it by 16x
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    cfg = pipeline.make_config(layers=layers)
    ckpt = pipeline.synthetic_checkpoint(cfg, seed=0)
    ocfg = oenc.Config(cfg)
    geom = opre.Geometry()
    feat_list = [dict(cfg['online']['input'], channel=0), dict(pipeline.BASELINE_FEAT, channel=0),
                 opre.get_feat_config('linear', 0), opre.get_feat_config('phase', 0),
                 opre.get_feat_config('linear', 1), opre.get_feat_config('phase', 1)]
    lengths, wavs = synth.fast_batch(batch, 160000, seed=1)

    def one():
        with torch.no_grad():
            f = opre.forward(wavs, feat_list, geom)
            hid = oenc.encoder_forward(f[0], ckpt['Transformer'], ocfg, lengths=torch.full((batch,), f[0].shape[1]))
            pred, res = oheads.spec_head(hid, ckpt['SpecHead'], ocfg, log=True)
            wav = odec.decode_wav(pred, f[3], lengths, geom, wavs[:, 1])
            loss = oobj.l1(res['log_predicted'], f[4], odec.get_length_masks(lengths // 160 + 1))
        return wav, loss

    # protocol of BASELINE.md section 3: 3 warm-ups, >= 10 timed iterations where the budget allows (a bounded sample: 10-30 s of CPU work)
    t0 = time.perf_counter()
    one()
    warm = time.perf_counter() - t0
    n_warm = 3 if warm * 14 < seconds_budget * 1.5 else 1
    for _ in range(n_warm - 1):
        one()
    iters = max(3, min(10, int(seconds_budget / max(warm, 1e-3)) - n_warm))
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    dt = (time.perf_counter() - t0) / iters
    cpu_name = ''
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    cpu_name = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return {'value': batch / dt, 'unit': 'utt/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n_warm} warm-ups + {iters} timed x batch of {batch} synthetic 10 s utterances, same pass (fp32 torch CPU oracle, L={layers}), '
                      f'{dt:.2f} s per batch on {cpu_name}'}


PMC_FILE = os.path.join(ROOT, 'profiles', 'r02_pmc_fetch_write.json')


# every kernel the prof family 'gemm_bf16' times (csrc/gemm*.hip): the traffic figure is the launch-weighted mean over the same launches
GEMM_KERNELS = ('gemm2_bf16_kernel', 'gemm3_bf16_kernel', 'gemm4_res_ln_kernel', 'gemm6_bf16_kernel', 'gemm6p_bf16_kernel', 'gemm7_res_ln_kernel',
                'gemm8_res24_ln_kernel', 'gemm5_bf16_kernel')


def pmc_traffic(substrings):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes of this same command (separate
    `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs, KB per dispatch; bench.py cannot run counter passes on itself).
    gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE tallies 128-B requests at 64 B for 16-B-per-lane
    streams, so it is doubled; WRITE_SIZE is exact.  Returns None when the file is absent."""
    try:
        with open(PMC_FILE) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        return None
    tot, n = 0.0, 0
    for name, v in d.items():
        if any(sub in name for sub in substrings) and 'FETCH_SIZE_KB_mean' in v and 'WRITE_SIZE_KB_mean' in v:
            tot += (2.0 * v['FETCH_SIZE_KB_mean'] + v['WRITE_SIZE_KB_mean']) * 1024.0 * v['launches']
            n += v['launches']
    return tot / n if n else None


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside torchrun: start N ranks as a CHILD `python -m torch.distributed.run` (this process has not
    touched the GPU and never execs), relay rank 0's JSON line and the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f'bench.py: the {n}-rank child exited with code {proc.returncode}' + ('' if line else ' and printed no JSON line'), file=sys.stderr)
        sys.exit(proc.returncode or 1)
    got = json.loads(line).get('n_gpus')
    if got != n:
        print(f'bench.py: asked for {n} ranks, the job observed {got}', file=sys.stderr)
        sys.exit(1)
    print(line, flush=True)
    sys.exit(0)


def launch_check(args):
    """--launch-check: the rank plumbing alone (process group, barrier, rank count by all-reduce, one JSON line), no kernels --
    what the CPU test of the launcher runs over gloo."""
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    dist.init_process_group(args.backend)
    dist.barrier()
    ones = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(ones)
    world = int(ones.item())
    rank = dist.get_rank()
    dist.barrier()
    dist.destroy_process_group()
    if world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but {world} ranks joined', file=sys.stderr)
        sys.exit(1)
    if rank == 0:
        print(json.dumps({'metric': 'launch-check', 'n_gpus': world, 'world_observed': world, 'backend': args.backend}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='utterances per GPU per step (configs[1]: 32)')
    ap.add_argument('--layers', type=int, default=6, help='encoder depth (6 = config/pretrain_sample.yaml; 3 = "base")')
    ap.add_argument('--workload', choices=('enhance', 'finetune', 'lstm'), default='enhance',
                    help="enhance = configs[1] (the headline metric); finetune = configs[3]'s Mockingjay training step "
                         '(fwd + L1 + bwd + gradient all-reduce + clip + BertAdam); lstm = the same step for the 3 x BiLSTM-256 head of '
                         'pseudo_noise.yaml:50-53 on raw features (run_active.sh); both reported as side measurements')
    ap.add_argument('--streams', type=int, default=1, help='process the batch as this many sub-batches on separate HIP streams (enhance workload)')
    ap.add_argument('--graph', action='store_true', help='replay the enhance pass as one hipGraph launch (serving-size batches are launch-bound)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-host-fed', action='store_true', help='skip the PCIe-inclusive side measurement')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend of the ranks ('nccl' = RCCL over xGMI)")
    ap.add_argument('--one-device', action='store_true', help='rehearsal: every rank uses cuda:0 (several ranks on a one-GPU box, e.g. --gpus 2 --backend gloo)')
    ap.add_argument('--launch-check', action='store_true', help='rank plumbing only (no kernels): used by the CPU test of the launcher')
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])            # never returns; nothing above touched the GPU
    if args.launch_check:
        return launch_check(args)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    distributed = world > 1 or 'RANK' in os.environ      # under torchrun always go through RCCL (also at N = 1)
    if world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}', file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print('bench.py needs an MI355X: the hot path has no CPU fallback', file=sys.stderr)
        sys.exit(2)
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # RCCL prints a version banner on stdout at communicator creation; the contract is ONE JSON line on stdout, so
    # everything before the final print goes to stderr at the file-descriptor level.
    sys.stdout.flush()
    saved_stdout_fd = os.dup(1)
    os.dup2(2, 1)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(args.backend, device_id=dev)
        dist.barrier()                      # creates the communicator now (not inside the timed region)
        ones = torch.ones(1, device=dev, dtype=torch.float64)
        dist.all_reduce(ones)               # the world size the collective itself sees
        world_observed = int(ones.item())
        if world_observed != args.gpus:
            print(f'bench.py: --gpus {args.gpus} but the all-reduce counted {world_observed} ranks', file=sys.stderr)
            sys.exit(2)
    else:
        world_observed = 1

    from speech_enhancement_by_s3prl_amd import _lib, pipeline, synth
    lib = _lib.load()
    cfg = pipeline.make_config(layers=args.layers)
    ckpt = pipeline.synthetic_checkpoint(cfg, seed=0)
    upstream = pipeline.build_upstream(ckpt, dev)
    pre = pipeline.build_preprocessor(cfg, dev)
    step = pipeline.UpstreamEnhanceStep(pre, upstream, streams=args.streams)
    if args.workload == 'finetune':
        import warnings
        warnings.simplefilter('ignore')
        from speech_enhancement_by_s3prl_amd.solver import get_optimizer
        del upstream
        model = pipeline.build_mockingjay(ckpt, dev)
        opt = get_optimizer(list(model.named_parameters()), lr=4e-5, warmup_proportion=0.07, training_steps=100000)
        ft = pipeline.MockingjayFinetuneStep(pre, model, opt)

        def step(wavs, lengths, max_len):          # same call shape as the enhance step
            loss, gn, skipped = ft(wavs, lengths)
            return loss.reshape(1), loss, None
    if args.workload == 'lstm':
        from speech_enhancement_by_s3prl_amd.lstm import LSTM
        from speech_enhancement_by_s3prl_amd.solver import get_optimizer
        del upstream
        head = LSTM(input_size=120, output_size=201, hidden_size=256, num_layers=3, bidirectional=True).to(dev)
        opt = get_optimizer(list(head.named_parameters()), lr=4e-5, warmup_proportion=0.07, training_steps=100000)
        ht = pipeline.HeadFinetuneStep(pre, head, opt)

        def step(wavs, lengths, max_len):
            loss, gn, skipped = ht(wavs, lengths)
            return loss.reshape(1), loss, None
    lengths, wavs = synth.fast_batch(args.batch, 160000, seed=1000 + rank, device=dev)   # resident in HBM before timing
    max_len = 160000
    if args.graph and args.workload == 'enhance':
        eager = step
        graphed = pipeline.GraphedStep(eager, wavs, lengths, max_len)

        def step(wavs, lengths, max_len):
            return graphed(wavs, lengths)

    def sync_all():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(wavs, lengths, max_len)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wav_pred, loss, _ = step(wavs, lengths, max_len)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        dist.barrier()
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert torch.isfinite(wav_pred).all() and torch.isfinite(loss)

    total_utts = args.batch * world * args.steps
    out = {
        'metric': {'enhance': 'enhanced 10s utts/sec', 'finetune': 'fine-tuned 10s utts/sec (Mockingjay training step)',
                   'lstm': 'trained 10s utts/sec (LSTM head training step)'}[args.workload], 'value': total_utts / elapsed, 'unit': 'utt/s', 'n_gpus': world_observed,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1000.0 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': 'configs[1]: TERA/Mockingjay upstream (6x768x12x3072, pretrain_sample.yaml) + '
                               'TransformerSpecPredictionHead, evaluate()-style pass, 10 s @ 16 kHz',
                   'batch_per_gpu': args.batch, 'global_batch': args.batch * world, 'layers': args.layers,
                   'frames': 1001, 'parallelism': f'dp{world} (utterance-sharded inference, no data-path collective)',
                   'weights': 'seeded random, real sizes',
                   'ranks': f'{world_observed} ranks counted by all-reduce over {args.backend}' if distributed else 'single process, no process group'},
    }
    if args.graph and args.workload == 'enhance':
        out['config']['launch'] = 'one hipGraph replay per step'
    if args.streams > 1 and args.workload == 'enhance':
        out['config']['streams'] = f'{args.streams} sub-batches of the batch in flight on separate HIP streams'
    if args.workload == 'lstm':
        out['config']['workload'] = ('config 5 style: 3 x BiLSTM-256 + Linear(512->201) head (pseudo_noise.yaml:50-53, 4.0 M params) on mel/log/delta-2 '
                                     'features: STFT/features, forward, masked log-L1, backward (BPTT), gradient all-reduce, clip 1.0, BertAdam')
        out['config']['parallelism'] = f'dp{world} (replicated parameters, one gradient all-reduce per step)'
    if args.workload == 'finetune':
        out['config']['workload'] = ('configs[3]: Mockingjay fine-tune step (6x768x12x3072 encoder + spec head, 43 M params): STFT/features, '
                                     'forward (train mode, dropout 0.1), masked log-L1, backward, flat-buffer gradient all-reduce, clip 1.0, BertAdam')
        out['config']['parallelism'] = f'dp{world} (replicated parameters, one gradient all-reduce per step)'

    collective_step = distributed and args.workload in ('finetune', 'lstm')      # the training steps all-reduce: every rank has to take part
    if not args.no_roofline and (rank == 0 or collective_step):
        # roofline leg: the same K steps with HIP events recorded (in-library, on the launch stream) around every
        # kernel of the dominant families.  Dominant kernel = the bf16 MFMA GEMM (QKV / out-proj / FFN / head).
        lib.se_prof_reset()
        lib.se_prof_enable(1)
        for _ in range(args.steps):
            step(wavs, lengths, max_len)
        torch.cuda.synchronize()
        lib.se_prof_enable(0)
    if rank == 0 and not args.no_roofline:
        fam = {}
        for kind, name in ((0, 'gemm_bf16'), (1, 'mhsa_fwd'), (2, 'stft'), (3, 'istft'), (6, 'mhsa_bwd')):
            ms, work, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
            lib.se_prof_read(kind, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n))
            fam[name] = (ms.value, work.value, n.value)
        g_ms, g_flop, g_n = fam['gemm_bf16']
        achieved = g_flop / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        # the committed PMC passes are of the default command: report them only when this run is that command
        pmc_matches = (args.workload == 'enhance' and args.batch == 32 and args.layers == 6 and args.streams == 1 and not args.graph)
        traffic = pmc_traffic(GEMM_KERNELS) if pmc_matches else None
        out['roofline'] = {'bound': 'mfma', 'kernel': 'gemm_bf16_kernel', 'achieved': achieved, 'peak': MFMA_BF16_PEAK_TFLOPS,
                           'unit': 'TFLOP/s', 'frac': achieved / MFMA_BF16_PEAK_TFLOPS, 'traffic': traffic,
                           'traffic_unit': 'HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, committed PMC passes: profiles/' + os.path.basename(PMC_FILE) + ')',
                           'launches': g_n, 'avg_launch_ms': g_ms / max(g_n, 1),
                           'algorithmic_flop_per_launch': g_flop / max(g_n, 1)}
        others = {}
        m_ms, m_flop, m_n = fam['mhsa_fwd']
        if m_ms > 0:
            a = m_flop / (m_ms * 1e-3) / 1e12
            others['mhsa_fwd_kernel'] = {'bound': 'mfma', 'achieved': a, 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                         'frac': a / MFMA_BF16_PEAK_TFLOPS, 'avg_launch_ms': m_ms / m_n, 'share_of_step_ms': m_ms / args.steps}
        b_ms, b_flop, b_n = fam['mhsa_bwd']
        if b_ms > 0:
            a = b_flop / (b_ms * 1e-3) / 1e12
            others['mhsa_bwd_kernels'] = {'bound': 'mfma', 'achieved': a, 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                          'frac': a / MFMA_BF16_PEAK_TFLOPS, 'avg_launch_ms': b_ms / b_n, 'share_of_step_ms': b_ms / args.steps}
        for name in ('stft', 'istft'):
            ms, byts, n = fam[name]
            if ms > 0:
                a = byts / (ms * 1e-3) / 1e9
                others[name + '_kernel'] = {'bound': 'hbm', 'achieved': a, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': a / HBM_PEAK_GBS,
                                            'avg_launch_ms': ms / n, 'share_of_step_ms': ms / args.steps}
        others['gemm_share_of_step_ms'] = g_ms / args.steps
        out['roofline_other_kernels'] = others

    if rank == 0 and args.workload == 'enhance' and not args.graph and not args.no_host_fed:
        # PCIe-inclusive side measurement (never `value`): the same K steps with every batch arriving from pinned HOST memory through the
        # double-buffered feeder (feeder.py; the reference copies synchronously on the compute stream, runner.py:431-432, 556-557)
        from speech_enhancement_by_s3prl_amd.feeder import HostBatchFeeder
        host_l, host_w = lengths.cpu().pin_memory(), wavs.cpu().pin_memory()
        feeder = HostBatchFeeder([(host_l, host_w)] * args.warmup, dev)
        for dl, dw in feeder:
            step(dw, dl, max_len)
        torch.cuda.synchronize()
        feeder.batches = [(host_l, host_w)] * args.steps          # same slots: device buffers are allocated once, as in a real loop
        t0 = time.perf_counter()
        for dl, dw in feeder:
            step(dw, dl, max_len)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out['host_fed'] = {'value': args.batch * args.steps / el, 'unit': 'utt/s', 'ms_per_step': 1000.0 * el / args.steps,
                           'bytes_per_step': host_w.numel() * 4 + host_l.numel() * 8,
                           'note': 'rank 0 only; batches DMA-ed from pinned host memory on a copy stream, double buffered (PCIe-inclusive; not the headline value)'}
        # the same with only the channels the pass reads (0 = noisy, 1 = clean; runner.py:558-561) crossing PCIe
        feeder = HostBatchFeeder([(host_l, host_w)] * args.warmup, dev, channels=2)
        for dl, dw in feeder:
            step(dw, dl, max_len)
        torch.cuda.synchronize()
        feeder.batches = [(host_l, host_w)] * args.steps
        t0 = time.perf_counter()
        for dl, dw in feeder:
            step(dw, dl, max_len)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out['host_fed']['two_channels'] = {'value': args.batch * args.steps / el, 'ms_per_step': 1000.0 * el / args.steps,
                                           'bytes_per_step': host_w.numel() * 4 * 2 // host_w.shape[1] + host_l.numel() * 8}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == 'enhance':
        try:
            out['cpu_baseline'] = cpu_baseline(batch=4, layers=args.layers, seconds_budget=28.0)
        except Exception as e:      # the baseline is a reported extra; never lose the GPU line
            out['cpu_baseline'] = {'value': None, 'unit': 'utt/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': f'failed: {e}'}

    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(saved_stdout_fd, 1)
    os.close(saved_stdout_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
