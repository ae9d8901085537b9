"""smnngp — MI355X-native scale-mixture NNGP engine.

Host-side mirror of the reference's API surface for ONE path (kernel build -> jittered Cholesky ->
log-marginal-likelihood / predictive mean+variance):

    smnngp.nt_kernels   get_mlp_kernel / get_dense_resnet_kernel / get_cnn_kernel   (experiments/nt_kernels.py)
    smnngp.predict      gradient_descent_mse_ensemble                               (neural_tangents.predict)
    smnngp.spax         kernels.NNGPKernel, likelihoods.*, models.SPR, utils, bijectors, base   (spax/*)

All arithmetic runs in libsmnngp.so (hand-written HIP for gfx950) through a ctypes C-ABI
(include/smnngp.h).  There is no CPU fallback: importing ``smnngp._lib`` raises if the library
has not been built (``python scale-mixtures-of-neural-network-gaussian-processes_amd/build.py``).
"""
__version__ = "0.1.0"
