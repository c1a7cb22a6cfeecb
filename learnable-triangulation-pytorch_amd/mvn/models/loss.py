"""Losses with the reference's names and signatures (mvn/models/loss.py of the reference).

The four keypoint losses act on (B, J, 3) tensors -- a few hundred floats -- and stay the reference's plain tensor expressions
("pure torch, reusable as is"; there is nothing for a kernel to win).  ``VolumetricCELoss`` is the one
that costs the reference real time -- a Python loop over samples with a ``.cpu()`` synchronisation each and a loop over joints
(loss.py:61-77) over a (B, V^3) distance volume it materialises per sample -- and is ONE liblt_hip launch here
(lt_volumetric_ce_fwd: nearest voxel, -log p and the sparse gradient, one workgroup per (sample, joint), no host round trip)."""
import torch
from torch import nn

import lt_hip as H
from mvn.utils import op


class KeypointsMSELoss(nn.Module):
    def forward(self, keypoints_pred, keypoints_gt, keypoints_binary_validity):
        dimension = keypoints_pred.shape[-1]
        loss = torch.sum((keypoints_gt - keypoints_pred) ** 2 * keypoints_binary_validity)
        return loss / (dimension * torch.clamp(torch.sum(keypoints_binary_validity), min=1))


class KeypointsMSESmoothLoss(nn.Module):
    def __init__(self, threshold=400):
        super().__init__()
        self.threshold = threshold

    def forward(self, keypoints_pred, keypoints_gt, keypoints_binary_validity):
        dimension = keypoints_pred.shape[-1]
        diff = (keypoints_gt - keypoints_pred) ** 2 * keypoints_binary_validity
        diff = torch.where(diff > self.threshold, torch.pow(diff.clamp(min=1e-30), 0.1) * (self.threshold ** 0.9), diff)
        return torch.sum(diff) / (dimension * torch.clamp(torch.sum(keypoints_binary_validity), min=1))


class KeypointsMAELoss(nn.Module):
    def forward(self, keypoints_pred, keypoints_gt, keypoints_binary_validity):
        dimension = keypoints_pred.shape[-1]
        loss = torch.sum(torch.abs(keypoints_gt - keypoints_pred) * keypoints_binary_validity)
        return loss / (dimension * torch.clamp(torch.sum(keypoints_binary_validity), min=1))


class KeypointsL2Loss(nn.Module):
    def forward(self, keypoints_pred, keypoints_gt, keypoints_binary_validity):
        loss = torch.sum(torch.sqrt(torch.sum((keypoints_gt - keypoints_pred) ** 2 * keypoints_binary_validity, dim=2)))
        return loss / torch.clamp(torch.sum(keypoints_binary_validity), min=1)


class _VolumetricCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coord_volumes, volumes, keypoints_gt, validity):
        B, J = volumes.shape[:2]
        nvox = volumes[0, 0].numel()
        dev = volumes.device
        cv = coord_volumes.float().contiguous()
        pr = volumes.float().contiguous()
        terms = torch.empty(B, J, dtype=torch.float32, device=dev)
        idx = torch.empty(B, J, dtype=torch.int32, device=dev)
        gval = torch.empty(B, J, dtype=torch.float32, device=dev)
        gt = keypoints_gt.float().contiguous()
        val = validity.float().reshape(B, J).contiguous()
        H.check(H.lib().lt_volumetric_ce_fwd(cv.data_ptr(), pr.data_ptr(), gt.data_ptr(), val.data_ptr(), terms.data_ptr(), idx.data_ptr(),
                                             gval.data_ptr(), B, J, nvox, H.cur_stream()), "lt_volumetric_ce_fwd")
        ctx.save_for_backward(idx, gval)
        ctx.volumes = volumes if volumes.requires_grad else None     # only its grad_fn / shape are used in backward
        return terms.sum() / (B * J)

    @staticmethod
    def backward(ctx, g):
        idx, gval = ctx.saved_tensors
        if ctx.volumes is None:
            return None, None, None, None
        # the gradient on the probabilities is one voxel per (sample, joint): it goes to lt_softargmax3d_bwd in sparse form
        # (a dense (B,J,V^3) gradient would be 570 MB at 32 samples)
        return None, op.sparse_prob_grad(ctx.volumes, idx, (gval * g).contiguous()), None, None


class VolumetricCELoss(nn.Module):
    """loss.py:52-80: mean over ALL (sample, joint) pairs of validity * -log(p[nearest voxel to the ground truth] + 1e-6)."""

    def forward(self, coord_volumes_batch, volumes_batch_pred, keypoints_gt, keypoints_binary_validity):
        H.require_gpu(volumes_batch_pred, "volumes_batch_pred")
        return _VolumetricCEFn.apply(coord_volumes_batch, volumes_batch_pred, keypoints_gt, keypoints_binary_validity)
