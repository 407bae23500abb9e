"""`sparseconvnet` operator API on the MI355X backend (hot-path subset).

    import occuseg_amd
    occuseg_amd.install()          # makes `import sparseconvnet as scn` resolve here
    import sparseconvnet as scn

Module names and constructor signatures follow sparseconvnet/__init__.py:9-37 of the
reference; everything outside the ScanNet UNet path raises NotImplementedError.
"""
forward_pass_multiplyAdd_count = 0
forward_pass_hidden_states = 0

from .. import SCN  # noqa: E402,F401
from .containers import (AddTable, ConcatTable, Identity, JoinConcatTable, JoinTable, NetworkInNetwork,  # noqa: E402,F401
                         ResidualConcatTable, Sequential)
from .conv import Convolution, Deconvolution, SubmanifoldConvolution, ValidConvolution  # noqa: E402,F401
from .io import InputLayer, OutputLayer  # noqa: E402,F401
from .norm import BatchNormalization, BatchNormLeakyReLU, BatchNormReLU  # noqa: E402,F401
from .tensor import Metadata, SparseConvNetTensor, optionalTensor, optionalTensorReturn, toLongTensor  # noqa: E402,F401
from .unet import UNet  # noqa: E402,F401
from .utils import add_feature_planes, upsample_feature  # noqa: E402,F401

_OUT_OF_SCOPE = (
    "Tanh Sigmoid ReLU LeakyReLU ELU SELU BatchNormELU AveragePooling MeanOnlyBNLeakyReLU "
    "ClassificationTrainValidate DenseToSparse Dropout BatchwiseDropout FullConvolution TransposeConvolution "
    "InputBatch BLInputLayer BLOutputLayer InputLayerInput MaxPooling PermutohedralSubmanifoldConvolution "
    "RandomizedStrideConvolution RandomizedStrideMaxPooling SparseToDense Sparsify spectral_norm UnPooling "
    "ShapeContext MultiscaleShapeContext SparseVggNet SparseResNet FullyConvolutionalNet").split()


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        raise NotImplementedError("sparseconvnet.%s is not on the ScanNet UNet hot path this backend provides "
                                  "(SURVEY.md section 8)" % name)
    raise AttributeError(name)
