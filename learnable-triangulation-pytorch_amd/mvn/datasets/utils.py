"""Batch plumbing either side of the hot path, with the reference's names and contracts
(mvn/datasets/utils.py of the reference): ``make_collate_fn`` builds the ``batch`` dict the models consume
(SURVEY.md section 8b), ``prepare_batch`` turns it into device tensors.

MI355X-side difference: the reference uploads the images view by view (B small H2D copies plus a device-side stack,
datasets/utils.py:47-53); here the whole (B, NV, H, W, 3) block goes through ONE pinned staging buffer and one asynchronous
copy, and the HWC -> CHW permutation happens on the device."""
import numpy as np
import torch


def make_collate_fn(randomize_n_views=True, min_n_views=10, max_n_views=31):
    """Reference datasets/utils.py:6-39: drops ``None`` items, optionally sub-samples the views (np.random, like the
    reference), and stacks per-view lists into the batch dict."""

    def collate_fn(items):
        items = [x for x in items if x is not None]
        if len(items) == 0:
            print("All items in batch are None")
            return None
        batch = dict()
        total_n_views = min(len(item["images"]) for item in items)
        indexes = np.arange(total_n_views)
        if randomize_n_views:
            n_views = np.random.randint(min_n_views, min(total_n_views, max_n_views) + 1)
            indexes = np.random.choice(np.arange(total_n_views), size=n_views, replace=False)
        batch["images"] = np.stack([np.stack([item["images"][i] for item in items], axis=0) for i in indexes], axis=0).swapaxes(0, 1)
        batch["detections"] = np.array([[item["detections"][i] for item in items] for i in indexes]).swapaxes(0, 1)
        batch["cameras"] = [[item["cameras"][i] for item in items] for i in indexes]
        batch["keypoints_3d"] = [item["keypoints_3d"] for item in items]
        batch["indexes"] = [item["indexes"] for item in items]
        try:
            batch["pred_keypoints_3d"] = np.array([item["pred_keypoints_3d"] for item in items])
        except Exception:       # the reference swallows a missing key the same way (:33-36)
            pass
        return batch

    return collate_fn


def worker_init_fn(worker_id):
    np.random.seed(np.random.get_state()[1][0] + worker_id)


STAGE_RING = 3          # pinned staging blocks per shape: prepare_batch of batch i+1, i+2 may fill while batch i's H2D copy is still queued
_staging = {}


def _pinned(shape, dtype):
    """Next slot of the pinned staging ring for this shape: [tensor, event of the last H2D copy that read it].  The caller waits for that
    event before overwriting the block (a loop that never synchronises -- inference without a loss.item() -- would otherwise hand batch
    i+1's pixels to batch i's queued copy) and records a new one behind its own copy.  Same hazard handling as the geometry ring of
    VolumetricTriangulationNet (GEO_RING) and the job table of lt_train.Adam."""
    key = (tuple(shape), dtype)
    ring = _staging.get(key)
    if ring is None:
        if len(_staging) > 4:
            _staging.clear()
        pin = torch.cuda.is_available()
        ring = _staging[key] = {"slots": [[torch.empty(shape, dtype=dtype).pin_memory() if pin else torch.empty(shape, dtype=dtype), None]
                                          for _ in range(STAGE_RING)], "next": 0}
    slot = ring["slots"][ring["next"]]
    ring["next"] = (ring["next"] + 1) % STAGE_RING
    return slot


def prepare_batch(batch, device, config=None, is_train=True):
    """batch dict -> (images (B,NV,3,H,W) fp32, keypoints_3d_gt (B,J,3), keypoints_3d_validity_gt (B,J,1),
    proj_matricies (B,NV,3,4) fp32), all on ``device`` -- reference datasets/utils.py:45-65."""
    device = torch.device(device)
    images = np.asarray(batch["images"])                     # (B, NV, H, W, 3), any real dtype
    if device.type == "cuda":
        slot = _pinned(images.shape, torch.float32)
        stage, ev = slot
        if ev is not None:
            ev.synchronize()                                          # the H2D copy that last read this block (STAGE_RING batches ago) has completed
        stage.copy_(torch.from_numpy(np.ascontiguousarray(images)))   # dtype conversion on the way into the pinned block
        dev = stage.to(device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(device))
    else:
        dev = torch.from_numpy(np.ascontiguousarray(images)).float()
    images_batch = dev.permute(0, 1, 4, 2, 3).contiguous()   # BxNVxHxWxC -> BxNVxCxHxW (reference img.py:95-98 per view)
    kp = np.stack(batch["keypoints_3d"], axis=0)
    keypoints_3d_batch_gt = torch.from_numpy(kp[:, :, :3]).float().to(device)
    keypoints_3d_validity_batch_gt = torch.from_numpy(kp[:, :, 3:]).float().to(device)
    proj = np.stack([np.stack([camera.projection for camera in camera_batch], axis=0) for camera_batch in batch["cameras"]], axis=0)
    proj_matricies_batch = torch.from_numpy(np.ascontiguousarray(proj.swapaxes(0, 1))).float().to(device)   # (B, NV, 3, 4)
    return images_batch, keypoints_3d_batch_gt, keypoints_3d_validity_batch_gt, proj_matricies_batch
