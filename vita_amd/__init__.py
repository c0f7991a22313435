"""vita_amd — MI355X-native (gfx950) hot path of VITA omni-modal inference.
The compute lives in vita_amd/lib/libvita_hip.so (hand-written HIP, C ABI in include/vita_hip.h);
this package is the host-side mirror of the reference's `vita.model` interface."""
__version__ = "0.1.0"
