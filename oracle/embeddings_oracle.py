"""CPU ORACLE for the precomputed-embedding loader.  TEST INFRASTRUCTURE ONLY (same rules as oracle/mpn_oracle.py).

Restates ``load_precomputed_embeddings`` (utils/rgb.py:150-188 of the reference) in numpy.  Pin: tests/golden/
g8_embedding_files.npz, produced by tools/make_golden.py from the imported reference function (its module-level imports of
skimage / torchvision / pycocotools / matplotlib, which the function does not use, are satisfied by empty modules)."""
import numpy as np


def load_precomputed_embeddings(stored, stored_frame, det_frame, det_id):
    """stored: what the per-frame files hold, concatenated in frame order ([n, 1 + D] or [n, 1 + C, H, W], id in column /
    channel 0); stored_frame [n]: the frame each row came from; det_frame / det_id: the ``frame`` and ``detection_id``
    columns of det_df.  Returns the selected embeddings without the id column / channel."""
    frames = np.unique(det_frame)                                   # rgb.py:169  frames_to_retrieve
    emb = stored[np.isin(stored_frame, frames)]                      # rgb.py:170-171 (only those files are opened)
    ids = emb[:, 0] if emb.ndim == 2 else emb[:, 0, 0, 0]            # rgb.py:177 / :184
    drop = np.asarray(sorted(set(ids.astype(np.int32).tolist()) - set(np.asarray(det_id).tolist())))   # :178 / :184
    emb = emb[~np.isin(ids, drop)]                                   # :179 / :185
    ids = emb[:, 0] if emb.ndim == 2 else emb[:, 0, 0, 0]
    assert (ids == np.asarray(det_id)).all(), "Problems loading embeddings. Indices between query and stored embeddings do not match. BOTH SHOULD BE SORTED!"
    return emb[:, 1:]                                                # :182 / :187
