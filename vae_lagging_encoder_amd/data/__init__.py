"""Input pipeline of the reference: corpus reading / vocabulary / equal-length batching for the text
scripts (reference data/text_data.py), the Omniglot tensor triple for image.py."""
from .text_data import MonoTextData, VocabEntry
from .image_data import ShuffledLoader, load_image_triple, sampler_order

__all__ = ["MonoTextData", "VocabEntry", "load_image_triple", "ShuffledLoader", "sampler_order"]
