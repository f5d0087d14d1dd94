from .api import _NoCodegen
CUDAConverter = _NoCodegen
