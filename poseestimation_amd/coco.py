"""Person rows -> COCO keypoint result records (evaluate.py:44-74).

The immediate consumer of the decode: for every person row the 17 COCO keypoints are
looked up through ``cfg.coco_to_ours`` (cfgs/config_vgg19.py:48); a missing part stays
``(0, 0, 0)``, a found one becomes ``(candidate[id][0], candidate[id][1], 1)``; the
record score is the person's score sum ``person[-2]``.
"""
import numpy as np

from .config import coco_to_ours


def person_keypoints(candidate, subset):
    """(n_persons, 17, 3) float64 keypoint table and (n_persons,) scores, vectorised."""
    subset = np.asarray(subset, np.float64).reshape(-1, 20)
    candidate = np.asarray(candidate, np.float64).reshape(-1, 4)
    ids = subset[:, coco_to_ours].astype(np.int64)          # (n, 17)
    have = ids != -1
    kp = np.zeros(ids.shape + (3,))
    if candidate.shape[0]:
        safe = np.where(have, ids, 0)
        kp[..., 0] = np.where(have, candidate[safe, 0], 0.0)
        kp[..., 1] = np.where(have, candidate[safe, 1], 0.0)
    kp[..., 2] = have
    return kp, subset[:, 18].copy()


def coco_records(candidate, subset, image_id):
    """The list of dicts evaluate.py:69-74 appends for one image."""
    kp, score = person_keypoints(candidate, subset)
    return [{"image_id": image_id, "category_id": 1, "keypoints": list(kp[i].reshape(-1)),
             "score": score[i]} for i in range(len(score))]


def batch_records(results, image_ids):
    """Records of a whole decoded batch (``predict_decode.Results``), image by image."""
    out = []
    for b, image_id in enumerate(image_ids):
        out.extend(coco_records(results.candidate(b), results.subset(b), image_id))
    return out


def device_records(ctx, image_ids):
    """The same records for the batch a context has just decoded, with the keypoint rows
    formed on the device (``pd_coco_rows``): nothing but the 17 x 3 rows, the scores and the
    image index of every person crosses to the host."""
    kp, score, image = ctx.coco_rows()
    return [{"image_id": image_ids[int(image[i])], "category_id": 1, "keypoints": list(kp[i].reshape(-1)),
             "score": score[i]} for i in range(len(score))]
