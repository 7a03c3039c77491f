"""mdir_amd -- MI355X-native descriptor extraction and ranking (the hot path of
jenicek/mdir + cirtorch) behind the reference's operator API.  See DESIGN.md."""
__version__ = "0.1.0"
