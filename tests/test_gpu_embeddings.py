"""Precomputed-embedding files: mpntrackseg_amd.embeddings.load_precomputed_embeddings -- file read on
the host, id filter / order check / id-column removal / optional pooling on the device through the C ABI -- against the
fixture generated from the reference function (bit-exact: pure data movement) and against the numpy oracle."""
import numpy as np
import pytest
import torch

from mpntrackseg_amd import embeddings as E
from oracle import embeddings_oracle as EO

pytestmark = pytest.mark.gpu


class Frame(dict):
    """the two det_df columns the loader reads (a pandas DataFrame works the same way)"""


def write_stored(tmp_path, name, stored, stored_frame):
    d = tmp_path / "processed_data" / name
    d.mkdir(parents=True)
    for f in np.unique(stored_frame):
        torch.save(torch.from_numpy(stored[stored_frame == f]), str(d / ("%d.pt" % f)))


def test_reference_fixture_bit_exact(golden, tmp_path):
    z = golden("g8_embedding_files.npz")
    df = Frame(frame=z["det_frame"], detection_id=z["det_id"])
    info = {"seq_path": str(tmp_path)}
    for tag, dim in (("1d", "1D"), ("3d", "3D")):
        write_stored(tmp_path, "emb" + tag, z["stored_" + tag], z["stored_frame"])
        out = E.load_precomputed_embeddings(df, info, "emb" + tag, use_cuda=True, embedding_dim=dim)
        assert out.is_cuda and tuple(out.shape) == z["out_" + tag].shape
        assert np.array_equal(out.cpu().numpy(), z["out_" + tag])
    pooled = E.load_precomputed_embeddings(df, info, "emb3d", use_cuda=True, embedding_dim='3D', pooled=True)
    np.testing.assert_allclose(pooled.cpu().numpy(), z["out_3d"].mean(axis=(2, 3)), rtol=0, atol=1e-6)


def test_larger_sequence_against_oracle_and_writer_round_trip(tmp_path):
    rng = np.random.RandomState(3)
    frames = np.repeat(np.arange(1, 41), rng.randint(1, 30, size=40))
    n = frames.size
    ids = np.arange(n) * 3 + 7                      # ascending, not contiguous
    emb = rng.randn(n, 256).astype(np.float32)
    node = rng.randn(n, 8, 4, 2).astype(np.float32)
    E.write_frame_embeddings(str(tmp_path), "reid", frames, ids, emb)
    E.write_frame_embeddings(str(tmp_path), "node", frames, ids, node)
    keep = rng.rand(n) < 0.6
    keep[frames == 17] = False                      # a frame that drops out of the query entirely
    df = Frame(frame=frames[keep], detection_id=ids[keep])
    info = {"seq_path": str(tmp_path)}
    out = E.load_precomputed_embeddings(df, info, "reid", embedding_dim='1D')
    stored = np.concatenate([ids[:, None].astype(np.float32), emb], axis=1)
    ref = EO.load_precomputed_embeddings(stored, frames, frames[keep], ids[keep])
    assert np.array_equal(out.cpu().numpy(), ref) and np.array_equal(ref, emb[keep])
    out3 = E.load_precomputed_embeddings(df, info, "node", embedding_dim='3D')
    assert np.array_equal(out3.cpu().numpy(), node[keep])


def test_errors_match_the_reference(tmp_path):
    frames = np.array([1, 1, 2, 2, 2])
    ids = np.array([0, 1, 2, 3, 4])
    E.write_frame_embeddings(str(tmp_path), "reid", frames, ids, np.ones((5, 4), np.float32))
    info = {"seq_path": str(tmp_path)}
    with pytest.raises(AssertionError, match="BOTH SHOULD BE SORTED"):    # query not in stored order
        E.load_precomputed_embeddings(Frame(frame=frames[::-1], detection_id=ids[::-1]), info, "reid")
    with pytest.raises(AssertionError, match="BOTH SHOULD BE SORTED"):    # an id the files do not hold
        E.load_precomputed_embeddings(Frame(frame=np.array([1, 2]), detection_id=np.array([0, 9])), info, "reid")
    with pytest.raises(AssertionError):                                    # invalid embedding_dim (rgb.py:165)
        E.load_precomputed_embeddings(Frame(frame=frames, detection_id=ids), info, "reid", embedding_dim='2D')
    out = E.load_precomputed_embeddings(Frame(frame=np.array([2]), detection_id=np.array([3])), info, "reid")
    assert tuple(out.shape) == (1, 4)
