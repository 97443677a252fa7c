"""coda_amd: MI355X-native masked-diffusion sampler for CoDA (drop-in for CoDALanguageModel.diffusion_generate)."""
from .model_config import CoDAConfig
from .generation_utils import DLMGenerationConfig, DLMGenerationMixin, DLMModelOutput
from .modeling_coda import CoDALanguageModel, CoDAModel

__all__ = ["CoDAConfig", "CoDALanguageModel", "CoDAModel", "DLMGenerationConfig", "DLMGenerationMixin", "DLMModelOutput"]
__version__ = "0.1.0"
