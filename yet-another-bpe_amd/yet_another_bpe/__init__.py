"""MI355X-native byte-level BPE trainer with the API of DreamOneX/yet-another-bpe (src/yet_another_bpe/__init__.py:5-13)."""

__version__ = "0.1.0"

from yet_another_bpe.tokenizer import BBPETokenizer
from yet_another_bpe.trainer import BBPEModel, BBPETrainer, BBPETrainerConfig

__all__ = [
    "BBPETokenizer",
    "BBPETrainer",
    "BBPETrainerConfig",
    "BBPEModel",
]
