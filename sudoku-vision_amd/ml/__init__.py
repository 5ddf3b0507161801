from .model import DigitCNN, count_parameters  # noqa: F401
