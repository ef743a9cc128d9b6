"""Staleness detection for packed device blobs derived from a module's parameters.

Every module that caches a packed copy of its weights (the plan of ``include/cookietts_hip.h``) keys that cache on
``param_key(self)`` and registers ``hook_invalidate(self)``: ``nn.Module.load_state_dict`` on a PARENT recurses
through ``_load_from_state_dict`` and never calls a child's ``load_state_dict`` override, so an override alone
would keep serving the old blob (silently wrong audio).  The key also catches optimizer steps and in-place
updates (version counter) and tensors swapped by ``.to()`` / ``.half()`` (data pointer).  Writes through
``param.data`` bypass the version counter: call ``repack()`` after those.
"""
from __future__ import annotations

import itertools


def param_key(module):
    return tuple((t.data_ptr(), t._version, t.dtype) for t in itertools.chain(module.parameters(), module.buffers()))


def hook_invalidate(module):
    """Drop the packed cache whenever this module's tensors are loaded, directly or through any parent."""
    module.register_load_state_dict_post_hook(lambda m, incompatible_keys: m._invalidate())
