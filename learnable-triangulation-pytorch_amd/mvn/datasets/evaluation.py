"""MPJPE evaluation with the reference's formula and result layout (``Human36MMultiViewDataset.evaluate`` and
``evaluate_using_per_pose_error``, mvn/datasets/human36m.py:190-273 of the reference) -- the metric BASELINE.json's
"MPJPE vs ref" is defined by.  The dataset class itself (licensed Human3.6M images, cv2 cropping) is out of scope; this is its
evaluation half as a function of the label table, so a caller holding ``labels`` can score our predictions exactly the way
the reference would."""
import numpy as np


def per_pose_errors(keypoints_gt, keypoints_3d_predicted, root_index=6):
    """(N,J,3) x2 -> (absolute per-pose error (N,), pelvis-relative per-pose error (N,)) in the units of the inputs (mm):
    mean over joints of the Euclidean distance, human36m.py:255,263-266."""
    gt = np.asarray(keypoints_gt); pr = np.asarray(keypoints_3d_predicted)
    absolute = np.sqrt(((gt - pr) ** 2).sum(2)).mean(1)
    gt_rel = gt - gt[:, root_index:root_index + 1, :]
    pr_rel = pr - pr[:, root_index:root_index + 1, :]
    relative = np.sqrt(((gt_rel - pr_rel) ** 2).sum(2)).mean(1)
    return absolute, relative


def evaluate_using_per_pose_error(per_pose_error, action_idx, action_names, subject_idx, subject_names):
    """human36m.py:190-235: per subject ('Average' first), per action with the two trials 'X-1' / 'X-2' merged into 'X'."""
    action_idx = np.asarray(action_idx); subject_idx = np.asarray(subject_idx)

    def by_actions(mask=None):
        if mask is None:
            mask = np.ones_like(per_pose_error, dtype=bool)
        scores = {"Average": {"total_loss": per_pose_error[mask].sum(), "frame_count": np.count_nonzero(mask)}}
        for ai, name in enumerate(action_names):
            e = per_pose_error[(action_idx == ai) & mask]
            scores[name] = {"total_loss": e.sum(), "frame_count": len(e)}
        for base in [name[:-2] for name in action_names if name.endswith("-1")]:
            comb = {"total_loss": 0.0, "frame_count": 0}
            for trial in (1, 2):
                n = "%s-%d" % (base, trial)
                comb["total_loss"] += scores[n]["total_loss"]
                comb["frame_count"] += scores[n]["frame_count"]
                del scores[n]
            scores[base] = comb
        return {k: (float("nan") if v["frame_count"] == 0 else v["total_loss"] / v["frame_count"]) for k, v in scores.items()}

    out = {"Average": by_actions()}
    for si, name in enumerate(subject_names):
        out[name] = by_actions(subject_idx == si)
    return out


def evaluate(labels, keypoints_3d_predicted, num_keypoints=17, kind="mpii", split_by_subject=False, transfer_cmu_to_human36m=False,
             transfer_human36m_to_human36m=False):
    """labels: the reference's label dict (``labels['table']`` with 'keypoints' (N,J,3), 'action_idx', 'subject_idx';
    ``labels['action_names']``, ``labels['subject_names']``).  Returns (relative MPJPE averaged over everything, full result
    dict) exactly like human36m.py:237-273."""
    keypoints_gt = np.asarray(labels["table"]["keypoints"])[:, :num_keypoints]
    keypoints_3d_predicted = np.asarray(keypoints_3d_predicted)
    if keypoints_3d_predicted.shape != keypoints_gt.shape:
        raise ValueError("`keypoints_3d_predicted` shape should be %s, got %s" % (keypoints_gt.shape, keypoints_3d_predicted.shape))
    root_index = 6
    if transfer_cmu_to_human36m or transfer_human36m_to_human36m:
        human36m_joints = [10, 11, 15, 14, 1, 4]
        cmu_joints = [10, 11, 15, 14, 1, 4] if transfer_human36m_to_human36m else [10, 8, 9, 7, 14, 13]
        keypoints_gt = keypoints_gt[:, human36m_joints]
        keypoints_3d_predicted = keypoints_3d_predicted[:, cmu_joints]
        root_index = 0
    absolute, relative = per_pose_errors(keypoints_gt, keypoints_3d_predicted, root_index)
    t = labels["table"]
    args = (t["action_idx"], labels["action_names"], t["subject_idx"], labels["subject_names"])
    result = {"per_pose_error": evaluate_using_per_pose_error(absolute, *args),
              "per_pose_error_relative": evaluate_using_per_pose_error(relative, *args)}
    return result["per_pose_error_relative"]["Average"]["Average"], result
