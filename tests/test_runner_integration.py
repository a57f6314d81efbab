"""Drop-in check against the reference's OWN runner (build container only: needs /root/reference).

``BBDMRunner`` is constructed with ``bbdm_amd``'s model classes patched in place of the reference's, exactly as
INTEGRATION.md prescribes, and driven through everything that does not need a GPU: ``initialize_model`` (+ ``weights_init``
via the overridden ``apply``), optimizer / scheduler creation from ``get_parameters()``, EMA registration / update /
``apply_shadow`` / ``restore``, checkpoint save -> strict load round trip.  (Forward / sampling need the MI355X and are
covered by the ``-m gpu`` tests against the oracle.)  Missing third-party packages of this container are stubbed."""
import argparse
import os
import sys
import types

import pytest
import torch
import torch.nn as nn
import yaml

pytestmark = pytest.mark.reference
REF = "/root/reference"


def _stub_modules():
    def mod(name, **attrs):
        import importlib.machinery
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__dict__.update(attrs)
        sys.modules.setdefault(name, m)
        return sys.modules[name]

    class _Writer:
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def add_image(self, *a, **k): pass
        def close(self): pass

    tb = mod("torch.utils.tensorboard", SummaryWriter=_Writer)
    torch.utils.tensorboard = tb
    mod("torchsummary", summary=lambda *a, **k: None)
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Compose=lambda x: x, ToTensor=lambda: None, Resize=lambda *a, **k: None,
                        RandomHorizontalFlip=lambda *a, **k: None)
    tv.utils = mod("torchvision.utils", make_grid=lambda x, **k: x[0], save_image=lambda *a, **k: None)
    mod("cv2")
    oc = mod("omegaconf", OmegaConf=type("OmegaConf", (), {}))
    mod("omegaconf.listconfig", ListConfig=list)
    oc.dictconfig = mod("omegaconf.dictconfig", DictConfig=dict)
    pl = mod("pytorch_lightning", LightningModule=nn.Module)
    return pl


def _ns(c):
    ns = argparse.Namespace()
    for k, v in c.items():
        setattr(ns, k, _ns(v) if isinstance(v, dict) else v)
    return ns


def test_reference_runner_drives_bbdm_amd_model(tmp_path, monkeypatch):
    _stub_modules()
    monkeypatch.syspath_prepend(REF)
    # torch >= 2.7 dropped ReduceLROnPlateau(verbose=...), which the reference still passes (BBDMRunner.py:63)
    import torch.optim.lr_scheduler as sched
    orig = sched.ReduceLROnPlateau

    class _Plateau(orig):
        def __init__(self, *a, verbose=None, **k):
            super().__init__(*a, **k)

    monkeypatch.setattr(sched, "ReduceLROnPlateau", _Plateau)
    # the reference's `datasets/` is a namespace package and loses against the installed HuggingFace `datasets`
    for name in [k for k in sys.modules if k == "datasets" or k.startswith("datasets.")]:
        monkeypatch.delitem(sys.modules, name)
    pkg = types.ModuleType("datasets")
    pkg.__path__ = [os.path.join(REF, "datasets")]
    monkeypatch.setitem(sys.modules, "datasets", pkg)
    import bbdm_amd
    import runners.DiffusionBasedModelRunners.BBDMRunner as R
    monkeypatch.setattr(R, "BrownianBridgeModel", bbdm_amd.BrownianBridgeModel)
    monkeypatch.setattr(R, "LatentBrownianBridgeModel", bbdm_amd.LatentBrownianBridgeModel)

    cfg = yaml.load(open(os.path.join(REF, "configs", "Template-BBDM.yaml")), Loader=yaml.FullLoader)
    cfg["model"]["BB"]["params"]["UNetParams"].update(dict(image_size=16, model_channels=32, channel_mult=(1, 2),
                                                           attention_resolutions=(2,), num_head_channels=32))
    cfg["model"]["EMA"].update(dict(use_ema=True, start_ema_step=0, update_ema_interval=1))
    config = _ns(cfg)
    config.args = argparse.Namespace(result_path=str(tmp_path), sample_at_start=False, train=True, resume_model=None,
                                     resume_optim=None, sample_to_eval=False, save_top=False, max_epoch=None,
                                     max_steps=None)
    config.training.use_DDP = False
    config.training.device = [torch.device("cpu")]
    config.training.local_rank = 0

    torch.manual_seed(11)
    runner = R.BBDMRunner(config)
    net = runner.net
    assert isinstance(net, bbdm_amd.BrownianBridgeModel)
    # weights_init reached every Conv2d / Linear through the overridden apply(): zero_module convs are re-initialised,
    # the Conv1d proj_out of the attention block stays zero
    sd = net.state_dict()
    assert float(sd["denoise_fn.out.2.weight"].abs().max()) > 0
    proj = [k for k in sd if k.endswith("proj_out.weight")]
    assert proj and all(float(sd[k].abs().max()) == 0 for k in proj)
    # optimizer over get_parameters(), scheduler
    assert len(runner.optimizer) == 1 and len(runner.scheduler) == 1
    n_opt = sum(p.numel() for g in runner.optimizer[0].param_groups for p in g["params"])
    assert n_opt == sum(p.numel() for p in net.denoise_fn.parameters())
    # EMA: register / update / apply_shadow / restore work on our parameter names
    ema = runner.ema
    assert set(ema.shadow) == {k for k, p in net.named_parameters() if p.requires_grad}
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.01)
    ema.update(net, with_decay=True)
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    ema.apply_shadow(net)
    assert any(not torch.equal(before[k], p) for k, p in net.named_parameters())
    ema.restore(net)
    assert all(torch.equal(before[k], p) for k, p in net.named_parameters())
    # checkpoint round trip through the runner's own state-dict plumbing (strict load)
    model_states, optim_states = runner.get_checkpoint_states()
    path = tmp_path / "ckpt.pth"
    torch.save(model_states, path)
    config.model.model_load_path = str(path)
    config.model.optim_sche_load_path = None
    fresh = R.BBDMRunner(config)
    for k, v in net.state_dict().items():
        assert torch.equal(fresh.net.state_dict()[k], v), k
    assert fresh.global_step == runner.global_step
