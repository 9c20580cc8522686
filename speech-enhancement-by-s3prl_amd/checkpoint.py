"""Reading the reference's checkpoint files (rows 8f-5 / 8b of SURVEY.md) without executing anything from them.

A `states-*.ckpt` of the reference (`runner.py:129-151`) and an S3PRL upstream checkpoint (`model.py:98-108, 145-153`) are `torch.save`d dicts of
tensors, numbers, strings and nested dicts -- plus ONE object: `Settings.Paras`, the `argparse.Namespace` of the run that wrote the file
(`runner.py:136`; read back at `run_downstream.py:104, 206`).  torch's weights-only unpickler (the default of `torch.load` from 2.6 on) refuses that
class, so a plain `torch.load(path)` raises on every file the reference or S3PRL ever wrote.  `load_checkpoint` allow-lists exactly that class for
the process (see allow_reference_checkpoints): still a no-code loader (`weights_only=True`; a Namespace is rebuilt from its `__dict__`, no function of the file runs).
"""
import argparse

import torch


def allow_reference_checkpoints():
    """The reference's scripts call `torch.load(path, map_location='cpu')` themselves (`run_downstream.py:103, 127, 137, 181, 199`, `runner.py:123`,
    `model.py:98, 143`); under torch >= 2.6 that call refuses the `argparse.Namespace` every one of its checkpoints holds.  Importing any product
    module (the S3PRL import paths of `s3prl_compat/` included) registers that one class with torch's weights-only unpickler, process-wide, so that
    those call sites keep working unchanged under `run_downstream.py`.  Still nothing of a file is executed: a Namespace is a plain attribute bag."""
    torch.serialization.add_safe_globals([argparse.Namespace])


allow_reference_checkpoints()


def load_checkpoint(path, map_location='cpu'):
    """`torch.load(path, map_location)` as the reference calls it (`run_downstream.py:103, 202`, `runner.py:122`, `model.py:98, 148`), restricted to
    tensors / containers / numbers / strings + `argparse.Namespace`."""
    # (not the `torch.serialization.safe_globals` context manager: leaving it REMOVES the class again, also when it had been registered before --
    #  the process-wide registration below would be gone after the first call, and with it the reference's own torch.load call sites)
    allow_reference_checkpoints()
    return torch.load(path, map_location=map_location, weights_only=True)


def reference_paras(downstream='LSTM', **overrides):
    """An `argparse.Namespace` with the fields `run_downstream.py:28-83` defines, as `Runner.save_model` stores it under `Settings.Paras`
    (`runner.py:129-137`); `downstream` is the one field read back from it (`run_downstream.py:206`)."""
    fields = dict(resume=None, name=None, n_jobs=12, dev_num=500, upstream='transformer', ckpt='', dropout=None, upstream2='transformer', ckpt2='',
                  dropout2=None, pseudo_clean=False, pseudo_noise=False, downstream=downstream, dckpt='', objective='L1', from_waveform=False,
                  from_rawfeature=False, optim='BertAdam', config='config/vcb.yaml', expdir='result', seed=1337, cpu=False, wandb=False,
                  eval_init=False, no_metric=False, save_best=False, active_sampling=False, record_num=5, sampler_device=None,
                  active_layerid=None, n_iterate=None, sync_sampler=False, train_speech=None, train_noise=None, test_speech=None, test_noise=None,
                  test=False, test_gradient=False)
    fields.update(overrides)
    return argparse.Namespace(**fields)
