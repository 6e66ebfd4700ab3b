"""Path parity with reference modules/feature_volume.py."""
from .cost_volume import FastFeatureVolumeManager, FeatureVolumeManager  # noqa: F401
