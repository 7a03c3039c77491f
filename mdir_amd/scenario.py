"""Scenario plumbing of the eval surface: deep overlay of YAML dictionaries, path joins,
wall-clock laps.

``dict_deep_overlay`` follows ``mdir/external/daan/core/experiments.py:1-38``: dicts
merge recursively, ``key*`` replaces, ``key+`` appends, a dict over a list patches
by integer index, two plain lists refuse to merge, anything else is replaced.
``path_join`` follows ``mdir/external/daan/ml/tools.py:26-48`` for the cases the
scenario files use; ``StopWatch`` is ``mdir/tools/stats.py:47-67``.
"""
import os
import time


def _overlay_item(original, key, item, list_replace):
    if isinstance(key, str) and key.endswith("*"):
        original[key[:-1]] = item
    elif isinstance(key, str) and key.endswith("+"):
        original[key[:-1]] += item
    elif key not in original:
        original[key] = item
    else:
        original[key] = dict_deep_overlay(original[key], item, list_replace=list_replace)


def dict_deep_overlay(*data, list_replace=False):
    """Overlay dictionaries left to right (the left-most one is modified in place)."""
    if len(data) == 1:
        return data[0]
    if len(data) > 2:
        head = dict_deep_overlay(data[0], data[1], list_replace=list_replace)
        return dict_deep_overlay(head, *data[2:], list_replace=list_replace)
    original, overlay = data
    if isinstance(original, (list, tuple)) and isinstance(overlay, dict):
        for key, item in overlay.items():
            assert isinstance(key, int)
            original[key] = dict_deep_overlay(original[key], item)
        return original
    if not isinstance(original, type(overlay)):
        return overlay
    if isinstance(overlay, dict):
        for key, item in overlay.items():
            _overlay_item(original, key, item, list_replace)
        return original
    if isinstance(overlay, list) and not list_replace:
        raise ValueError("Cannot implicitly merge two lists, use key* or key+ when inheriting: "
                         "(list1: %s, list2: %s)" % (str(original), str(overlay)))
    return overlay


def validate_hash(content, path):
    """A name that ends in ``-<8+ hex digits>.<ext>`` promises the sha256 prefix of its content
    (``validate`` of ``mdir/tools/utils.py:27-34``; same error)."""
    import hashlib
    import re
    match = re.search(r'.*-([a-f0-9]{8,})\.[a-zA-Z0-9]{2,}$', path)
    if match:
        stored = match.group(1)
        computed = hashlib.sha256(content).hexdigest()[:len(stored)]
        if computed != stored:
            raise ValueError("Computed hash '%s' is not consistent with stored hash '%s'" % (computed, stored))


def resource_dirs():
    """Where a model / whitening file named by a URL is looked for locally: ``$MDIR_AMD_MODELS`` (``:``-separated),
    then ``<data root>/networks``."""
    from .datasets import get_data_root
    dirs = [d for d in os.environ.get("MDIR_AMD_MODELS", "").split(":") if d]
    return dirs + [os.path.join(get_data_root(), "networks")]


def open_resource(path):
    """Bytes of a checkpoint / whitening file.  The reference accepts URLs and downloads them (``load_url``,
    ``mdir/tools/utils.py:36-41``); nothing is fetched here (the MI355X boxes have no network): a URL is answered
    from a local copy of the same file NAME under ``resource_dirs()``, checked against the sha256 suffix of the name
    exactly as the reference checks its download, and otherwise with an error that says where to put the file."""
    import io
    if path.startswith("http://") or path.startswith("https://"):
        name = path.rstrip("/").rsplit("/", 1)[-1]
        for d in resource_dirs():
            local = os.path.join(d, name)
            if os.path.isfile(local):
                with open(local, "rb") as handle:
                    content = handle.read()
                validate_hash(content, path)
                return io.BytesIO(content)
        raise RuntimeError("'%s' is a URL and nothing is downloaded here: put '%s' into one of %s (or give the scenario "
                           "the local path)" % (path, name, resource_dirs()))
    with open(path, "rb") as handle:
        return io.BytesIO(handle.read())


def path_join(*paths):
    """Join, letting a later absolute path or URL win."""
    out = ""
    for p in paths:
        if not p:
            continue
        if p.startswith("/") or "://" in p or not out:
            out = p
        else:
            out = os.path.join(out, p)
    return out


class StopWatch:
    def __init__(self):
        self.timings = {}
        self.time0 = time.time()
        self.time_reset = self.time0

    def reset(self, include_total=True):
        timings, self.timings = self.timings, {}
        self.time0 = time.time()
        if include_total:
            timings["total_s"] = self.time0 - self.time_reset
        self.time_reset = self.time0
        return timings

    def lap(self, name):
        now = time.time()
        self.timings[name] = now - self.time0
        self.time0 = now
