"""GPU side of the dataset adjacency (mvn/datasets/utils.py): prepare_batch's pinned staging ring under an unsynchronised caller."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_prepare_batch_back_to_back_without_sync_keeps_batches_apart():
    """batches prepared back to back behind a busy GPU (no synchronisation in between) must each arrive with their
    OWN pixels -- the staging block is a ring guarded by events, not one buffer overwritten while its H2D copy is still queued."""
    from mvn.datasets import utils as du
    from mvn.utils.multiview import Camera
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    cam = Camera(np.eye(3), np.zeros((3, 1)), np.eye(3))

    def make(i):
        return {"images": rng.randint(0, 255, size=(2, 2, 64, 64, 3)).astype(np.uint8) + 0 * i,
                "keypoints_3d": [rng.randn(17, 4) for _ in range(2)], "cameras": [[cam, cam], [cam, cam]]}
    batches = [make(i) for i in range(2 * du.STAGE_RING + 1)]
    busy = torch.randn(4096, 4096, device=dev)
    outs = []
    for b in batches:
        for _ in range(3):
            busy = busy @ busy * 1e-4          # keep the stream busy so that the H2D copies queue up behind compute
        outs.append(du.prepare_batch(b, dev)[0])
    torch.cuda.synchronize()
    for b, o in zip(batches, outs):
        want = torch.from_numpy(b["images"]).float().permute(0, 1, 4, 2, 3)
        assert torch.equal(o.cpu(), want)
