"""exastencils_amd -- MI355X-native geometric multigrid hot path behind ExaStencils' kernel boundary.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/examg.h), and the host-side
mirror of the reference's ExaSlang-4 surface (layouts, fields, stencils, `loop over`,
`communicate`, `apply bc`, leveled cycle functions).  The compute path is libexamg.so; there is no
CPU fallback: importing `exastencils_amd.lib` without the built library, or running it without a
GPU, raises.
"""
__version__ = "0.1.0"
