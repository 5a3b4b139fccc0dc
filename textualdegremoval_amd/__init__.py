"""textualdegremoval_amd -- MI355X-native guided-restoration train step.

Host side mirrors the reference's operator surface (models/archs) and step API
(models/image_restoration_ref_model); device work is libtdr_hip.so (csrc/)."""
__version__ = '0.1.0'
