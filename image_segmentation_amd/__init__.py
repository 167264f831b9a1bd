"""image_segmentation_amd -- MI355X-native segmentation forward/backward hot path
(drop-in for the reference's unet/unet.py, clip/clipunet.py decoder, autoencoder/autoencoder.py family,
utils/weighted_loss.py losses and the utils/training.py loops).  Requires the in-tree HIP library (python -m image_segmentation_amd.build)."""
from .ops import set_compute_dtype, get_compute_dtype      # noqa: F401
from .unet import unet, DoubleConvReLU, Down, Up             # noqa: F401
from .losses import (CrossEntropyLoss, WeightedMemoryEfficientDiceLoss, WeightedDiceCELoss,     # noqa: F401
                     WeightedMemoryEfficientDiceLossPrompt, WeightedDiceNLLLoss)
from .clipunet import ClipUNet, UNetDecoder, DecoderBlock, ClipViTEncoder                   # noqa: F401
from .autoencoder import SegmentationAutoencoder, ReconstructionAutoencoder               # noqa: F401
from .prompt import PromptModel                                                            # noqa: F401
