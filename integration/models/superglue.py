"""``models.superglue`` as the reference's callers import it (``test.py:11``: ``from models.superglue import SuperGlue``).

The reference's ``SuperGlue`` (``models/superglue.py:315-625``) is the SuperGlue baseline: the same encoders, attentional GNN,
final projection and Sinkhorn layer as ``MDGAT`` with EVERY layer fully connected.  As shipped it is broken (its forward
passes ``self.k`` - never set - to a two-argument ``AttentionalGNN.forward``, ``superglue.py:418`` / ``267``);
functionally it is ``MDGAT`` with ``k = []``.  That is what this class is: same constructor config (a ``k`` entry is ignored),
same parameter names for ``descriptor == 'FPFH'`` (``kenc``, ``denc``, ``gnn.layers.*``, ``final_proj``, ``bin_score``), same
``forward(dict) -> dict``, running on the gfx950 library."""
from mdgat_matcher_amd.mdgat import MDGAT


class SuperGlue(MDGAT):
    def __init__(self, config):
        super().__init__({**config, 'k': []})


__all__ = ['SuperGlue']
