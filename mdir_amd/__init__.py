"""mdir_amd -- MI355X-native descriptor extraction and ranking (the hot path of
jenicek/mdir + cirtorch) behind the reference's operator API.  See DESIGN.md."""
import os

__version__ = "0.2.0"

# MIOpen's default find mode answers the FIRST convolution call of every new (layer, input shape) with a search whose
# candidates include its naive reference kernels: 4.2 s for one ResNet101 pass on a new image size from empty caches
# (73 % of the GPU time of a whole benchmark run in round 1).  FAST mode (2) takes the find-db / heuristic answer:
# 0.8 s for the same pass and the same steady state (tools/miopen_probe.sh, fresh user db and kernel cache per mode:
# default 4.25 s, NORMAL 4.11, FAST 0.78, HYBRID 4.17, naive solver switched off 1.29).  An image list has dozens of
# sizes, each at three scales, so this is what extraction wants; an explicit setting of the user wins.
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
# Round 5 (profiles/r05_miopen.md): on the 16 sizes x 3 scales of the extraction list FAST and a full find persisted to a user
# find-db give the same steady state (71.0 against 70.9 ms summed over the 48 shapes); the find costs 26 minutes of first calls.
# In immediate mode MIOpen WARNS for every layer shape whose preferred solver wants a workspace PyTorch does not hand over
# ("GetSolutionsFallback ... workspace required ... provided ptr: 0") and takes the next solver: hundreds of lines per image list.
# Errors stay visible; MIOPEN_LOG_LEVEL set by the user wins.
os.environ.setdefault("MIOPEN_LOG_LEVEL", "3")
