"""MI355X-native EINCM objective-and-gradient engine (host-side mirror of the reference interface).

The compute path is the HIP library built from ``csrc/`` (C-ABI in ``include/eincm.h``); this package
is the thin Python host layer over it.  There is no CPU fallback: importing the engine without the
built library raises.
"""
