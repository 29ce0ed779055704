from .diffusion_classifier import DiffusionClassifier  # noqa: F401
