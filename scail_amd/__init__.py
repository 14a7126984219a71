"""scail_amd -- MI355X (gfx950) native implementation of the SCAIL video-DiT sampling hot path.

The compute lives in ``libscail_hip.so`` (hand-written HIP kernels behind the C ABI declared in
``include/scail_hip.h``).  This package is the host side: it mirrors the reference's Python
plugin interface for that path (``DiffusionTransformer``, ``RFSampler``/``Denoiser``/
``VanillaCFG``/``OpenAIWrapper``, ``SATVideoDiffusionEngine.sample``) so the reference's configs
and call sites keep working, and uses PyTorch only for device memory, streams and
``torch.distributed`` (RCCL).

There is NO CPU fallback: importing ``scail_amd.lib`` without the built library, or calling an
op without a GPU, raises.
"""

__version__ = "0.1.0"
