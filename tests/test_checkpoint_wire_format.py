"""CPU test of the checkpoint boundary: files shaped like the reference's own -- `Settings.Paras` an `argparse.Namespace`
(`runner.py:129-137`), an S3PRL upstream checkpoint with `Transformer` / `SpecHead` / `Settings.Config` (`model.py:98-108, 145-153`) -- open through
the product's loader and construct `SpecHead`, `TRANSFORMER` and `Mockingjay` FROM THE PATH, as `run_downstream.py:167-217` does.  The plain
`torch.load` of the reference's call sites refuses such a file under the installed torch (weights-only default): that is what round 4 shipped."""
import argparse

import pytest
import torch


def _upstream_file(tmp_path):
    from speech_enhancement_by_s3prl_amd import pipeline
    cfg = pipeline.make_config(layers=1, hidden=256, heads=4, intermediate=512)
    ckpt = pipeline.synthetic_checkpoint(cfg, seed=3)
    assert isinstance(ckpt['Settings']['Paras'], argparse.Namespace)
    path = str(tmp_path / 'states-500000.ckpt')
    torch.save(ckpt, path)
    return path, ckpt


def test_reference_call_sites_keep_working_once_the_package_is_imported(tmp_path):
    """`torch.load(path, map_location='cpu')` as the reference's own scripts call it (`run_downstream.py:103, 127, 137, 181, 199`, `runner.py:123`):
    a fresh interpreter WITHOUT this package refuses the file (torch >= 2.6: weights-only default, argparse.Namespace not allow-listed); with any
    product module imported the same call opens it (checkpoint.allow_reference_checkpoints), still under the weights-only unpickler."""
    import subprocess
    import sys
    path, _ = _upstream_file(tmp_path)
    bare = subprocess.run([sys.executable, '-c', f"import torch; torch.load({path!r}, map_location='cpu')"], capture_output=True, text=True)
    assert bare.returncode != 0 and 'Namespace' in bare.stderr
    from speech_enhancement_by_s3prl_amd import _lib      # noqa: F401  (any product module)
    got = torch.load(path, map_location='cpu')
    assert got['Settings']['Paras'].downstream == 'SpecHead'
    got = torch.load(path, map_location='cpu', weights_only=True)
    assert got['Settings']['Paras'].seed == 1337


def test_load_checkpoint_opens_reference_shaped_files(tmp_path):
    from speech_enhancement_by_s3prl_amd.checkpoint import load_checkpoint, reference_paras
    path, ckpt = _upstream_file(tmp_path)
    got = load_checkpoint(path)
    assert sorted(got) == ['Settings', 'SpecHead', 'Transformer']
    assert got['Settings']['Paras'].downstream == 'SpecHead' and got['Settings']['Paras'].seed == 1337
    assert got['Settings']['Config'] == ckpt['Settings']['Config']
    for k, v in ckpt['Transformer'].items():
        assert torch.equal(got['Transformer'][k], v)
    # a downstream states-*.ckpt as Runner.save_model writes it (runner.py:129-151), optimizer state included
    lin = torch.nn.Linear(4, 3)
    opt = torch.optim.Adam(lin.parameters())
    lin(torch.randn(2, 4)).sum().backward()
    opt.step()
    states = {'Downstream': lin.state_dict(), 'Optimizer': opt.state_dict(), 'Global_step': 7,
              'Settings': {'Config': {'model': {'LSTM': {'hidden_size': 256}}}, 'Paras': reference_paras(downstream='LSTM', expdir='result/x')}}
    p2 = str(tmp_path / 'states-7.ckpt')
    torch.save(states, p2)
    back = load_checkpoint(p2)
    assert back['Global_step'] == 7 and back['Settings']['Paras'].expdir == 'result/x'
    # run_downstream.py:206: the model section is looked up by the Namespace's field
    assert back['Settings']['Config']['model'][back['Settings']['Paras'].downstream] == {'hidden_size': 256}
    assert torch.equal(back['Downstream']['weight'], lin.weight)


def test_load_checkpoint_still_executes_nothing(tmp_path):
    """anything beyond tensors / containers / Namespace stays refused: a pickled callable in the file must not load"""
    from speech_enhancement_by_s3prl_amd.checkpoint import load_checkpoint
    path = str(tmp_path / 'bad.ckpt')
    torch.save({'Settings': {'Paras': argparse.Namespace(x=1)}, 'fn': print}, path)
    with pytest.raises(Exception):
        load_checkpoint(path)


def test_modules_construct_from_the_path(tmp_path):
    """model.py:94-117 (SpecHead), model.py:129-153 (Mockingjay -> TRANSFORMER): construction from a checkpoint PATH needs no GPU"""
    from speech_enhancement_by_s3prl_amd.heads import Mockingjay, SpecHead
    from speech_enhancement_by_s3prl_amd.transformer import TRANSFORMER
    path, ckpt = _upstream_file(tmp_path)
    head = SpecHead(output_size=201, ckpt=path)
    assert torch.equal(head.spechead.output.weight.detach(), ckpt['SpecHead']['output.weight'])
    options = {'ckpt_file': path, 'load_pretrain': 'True', 'no_grad': 'True', 'dropout': 'default', 'spec_aug': 'False', 'spec_aug_prev': 'True',
               'weighted_sum': 'False', 'select_layer': -1, 'permute_input': 'False'}
    up = TRANSFORMER(options, inp_dim=80)
    sd = up.state_dict()
    key = next(k for k in sd if k.endswith('input_representations.spec_transform.weight'))
    assert torch.equal(sd[key].cpu(), ckpt['Transformer']['input_representations.spec_transform.weight'])
    mj = Mockingjay(dckpt=path)
    assert isinstance(mj.mockingjay, TRANSFORMER)
