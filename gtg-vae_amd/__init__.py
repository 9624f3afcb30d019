"""gtg-vae graph-VAE hot path on AMD MI355X (gfx950): hand-written HIP kernels
behind a C ABI (csrc/, include/gtgvae.h) and the reference's Python class
surface on top of it.

    from gtg_vae_amd import GraphVAE, GTVAE, mst_graph, mst_graph_batched
"""
from .maxspantree import mst_graph, mst_graph_batched  # noqa: F401
from .graph_vae import GraphVAE, VGAE_encoder, MLP_VAE_plain  # noqa: F401
from .gtvae import GTVAE  # noqa: F401
from ._lib import set_default_matrix_mode  # noqa: F401

__all__ = ["GraphVAE", "VGAE_encoder", "MLP_VAE_plain", "GTVAE", "mst_graph", "mst_graph_batched",
           "set_default_matrix_mode"]
