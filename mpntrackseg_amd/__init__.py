"""mpntrackseg_amd -- MI355X-native message-passing hot path of MPNTrackSeg.

``from mpntrackseg_amd.mpn import MOTMPNet, MetaLayer`` mirrors
``mot_neural_solver.models.mpn`` (reference); the numeric work happens in ``csrc/libmpnhip.so``.
"""
__all__ = ["capi", "mpn", "mlp", "synth"]
