"""fisher_rast: MI355X-native host layer over libfisher_rast.so (see include/fisher_rast.h)."""
from ._lib import FisherRastError, load, SO_PATH  # noqa: F401
