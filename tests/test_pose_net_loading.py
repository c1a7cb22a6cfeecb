"""CPU: SURVEY.md section 8 row A4 -- ``get_pose_net`` (reference mvn/models/pose_resnet.py:321-377): the pretrained-load
branch (``state_dict`` wrapper, ``module.`` prefix strip, shape filter, partial copy of a mismatched ``final_layer`` with
xavier/zero re-initialisation of the rest) and ``style == 'caffe'`` (Bottleneck_CAFFE at every depth, stride on the first
1x1).  Expectations are written out independently here; when /root/reference is present (build container only) the real
reference is run on the same fake checkpoint as a cross-check."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import ref_loader, spec, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(nl, style="simple", ckpt="", init=False, joints=17, alg=False, vol=False):
    return synth.AttrDict(num_layers=nl, style=style, num_joints=joints, alg_confidences=alg, vol_confidences=vol,
                          init_weights=init, checkpoint=ckpt)


def _fake_checkpoint(path, nl, joints_ckpt, wrap, prefix, seed=3):
    """A 'pretrained' backbone with ``joints_ckpt`` heatmap channels, every key prefixed with ``module.`` (a DataParallel
    save), optionally inside {'state_dict': ...}; plus one key the model does not have and one with a wrong shape."""
    sp = spec.pose_resnet_spec(nl, joints_ckpt, False, False, "")
    sd = synth.make_state_dict(sp, seed=seed, basic_block=(nl < 50))
    sd["not_in_model.weight"] = torch.ones(3)
    sd["layer1.0.conv1.weight"] = torch.ones(5, 5, 1, 1)          # wrong shape: must be filtered out, not raise
    ck = {prefix + k: v for k, v in sd.items()}
    torch.save({"state_dict": ck, "epoch": 3} if wrap else ck, path)
    return sd


@pytest.mark.parametrize("wrap,prefix,joints_ckpt", [(True, "module.", 21), (False, "", 11), (False, "module.", 17)])
def test_get_pose_net_pretrained_load(tmp_path, capsys, wrap, prefix, joints_ckpt):
    from mvn.models import pose_resnet
    nl, J = 18, 17
    path = str(tmp_path / "ckpt.pth")
    ck = _fake_checkpoint(path, nl, joints_ckpt, wrap, prefix)
    torch.manual_seed(1234)
    fresh = pose_resnet.get_pose_net(_cfg(nl), device="cpu").state_dict()       # ctor-default values of the same seed
    torch.manual_seed(1234)
    m = pose_resnet.get_pose_net(_cfg(nl, ckpt=path, init=True), device="cpu")
    out = capsys.readouterr().out
    assert "Loading pretrained weights from: " + path in out and "Successfully loaded pretrained weights for backbone" in out
    sd = m.state_dict()
    n = min(J, joints_ckpt)
    for k, v in sd.items():
        if k == "layer1.0.conv1.weight":                    # shape mismatch -> filtered, ctor value kept
            assert torch.equal(v, fresh[k]), k
        elif k.startswith("final_layer.") and joints_ckpt != J:
            assert torch.equal(v[:n], ck[k][:n]), k          # the first min(J, J_ckpt) filters are copied ...
            if k.endswith("bias"):
                assert torch.equal(v[n:], torch.zeros_like(v[n:]))   # ... the rest: zeros (bias)
            elif J > n:
                lim = float(np.sqrt(6.0 / (256 + J)))                # ... or xavier_uniform of the (J,256,1,1) weight
                assert float(v[n:].abs().max()) <= lim and float(v[n:].abs().max()) > 0.3 * lim
        elif k.endswith("num_batches_tracked"):
            assert int(v) == int(ck[k])
        else:
            assert torch.equal(v, ck[k]), k
    if joints_ckpt != J:
        assert "Reiniting final layer" in out
    assert "were not inited" in out and "not_in_model.weight" in out and "layer1.0.conv1.weight" in out
    if ref_loader.available():      # cross-check against the real reference on the same file and RNG state
        # (own process: the reference's package is called ``mvn`` too)
        dump = str(tmp_path / "ref_sd.pth")
        code = ("import sys, torch; sys.path.insert(0, %r); from oracle import ref_loader, synth; mvn = ref_loader.load(); "
                "torch.manual_seed(1234); cfg = synth.AttrDict(num_layers=%d, style='simple', num_joints=%d, alg_confidences=False, "
                "vol_confidences=False, init_weights=True, checkpoint=%r); "
                "torch.save(mvn.models.pose_resnet.get_pose_net(cfg, device='cpu').state_dict(), %r)") % (ROOT, nl, J, path, dump)
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH="")
        subprocess.run([sys.executable, "-c", code], check=True, env=env, capture_output=True, cwd=str(tmp_path))
        ref = torch.load(dump)
        assert list(ref) == list(sd)
        for k in sd:
            assert torch.equal(ref[k], sd[k]), k


def test_get_pose_net_caffe_style_structure():
    """'caffe': the block stride sits on conv1 (1x1) instead of conv2 (3x3); a basic-block depth still gets expansion-4
    bottlenecks (the reference swaps the block class for any num_layers)."""
    from mvn.models import pose_resnet
    for nl in (50, 18):
        m = pose_resnet.get_pose_net(_cfg(nl, style="caffe"), device="cpu")
        sp = spec.pose_resnet_spec(nl, 17, False, False, "", caffe=True)
        sd = m.state_dict()
        assert list(sd) == list(sp) and all(tuple(sd[k].shape) == sp[k][0] for k in sp)
        blk = m.layer2[0]
        assert blk.conv1.stride == (2, 2) and blk.conv2.stride == (1, 1) and blk.conv3.weight.shape[0] == 4 * blk.conv2.weight.shape[0]
        s = pose_resnet.get_pose_net(_cfg(50, style="simple"), device="cpu").layer2[0]
        assert s.conv1.stride == (1, 1) and s.conv2.stride == (2, 2)
    b = pose_resnet.get_pose_net(_cfg(18), device="cpu").layer2[0]
    assert not hasattr(b, "conv3") and b.conv1.stride == (2, 2)


def test_caffe_oracle_vs_reference_golden(golden_dir):
    from oracle import vol_oracle as O
    g = np.load(os.path.join(golden_dir, "nets_caffe.npz"))
    gen = torch.Generator().manual_seed(31)
    for nl, hw in ((50, 128), (18, 64)):
        sp = spec.pose_resnet_spec(nl, 17, False, False, "", caffe=True)
        assert len(sp) == int(g["rn%d_nkeys" % nl])
        sd = synth.make_state_dict(sp, seed=700 + nl)
        assert np.allclose(synth.state_dict_checksum(sd), g["rn%d_sd_digest" % nl], rtol=1e-12)
        x = torch.randn(2, 3, hw, hw, generator=gen)
        with torch.no_grad():
            hm, ft, _, _ = O.pose_resnet(sd, x, nl, prefix="", caffe=True)
        assert float((ft[:, :, ::2, ::2] - torch.from_numpy(g["rn%d_feat_s2" % nl])).abs().max()) <= 2e-5 * float(np.abs(g["rn%d_feat_s2" % nl]).max())
        assert float((hm - torch.from_numpy(g["rn%d_hm" % nl])).abs().max()) <= 2e-5 * float(np.abs(g["rn%d_hm" % nl]).max())
