"""Path parity with reference modules/mesh_hint_volume.py."""
from .cost_volume import FastFeatureMeshHintVolumeManager, FeatureMeshHintVolumeManager  # noqa: F401
