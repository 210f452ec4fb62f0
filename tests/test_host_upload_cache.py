"""CPU checks of the host layer's upload cache (penguin/jl_amd/api.py, _UploadCache): data re-evaluated at every step of a
host-driven loop are handed to the library only when their values changed."""
import numpy as np

from penguin.jl_amd.api import _UploadCache


def test_first_use_counts_as_changed_then_equal_values_do_not():
    c = _UploadCache()
    a, b = np.zeros(5), np.ones(5)
    assert c.changed("f", a, b)                       # nothing sent yet
    assert not c.changed("f", np.zeros(5), np.ones(5))   # other objects, same values
    assert not c.changed("f", a, b)


def test_a_change_of_any_array_any_entry_or_the_shape_is_seen():
    c = _UploadCache()
    a, b = np.zeros(5), np.ones(5)
    c.changed("f", a, b)
    b2 = b.copy(); b2[3] = 1.0 + 1e-16                # (rounds to 1.0: not a change)
    assert not c.changed("f", a, b2)
    b2[3] = np.nextafter(1.0, 2.0)
    assert c.changed("f", a, b2)                      # one ulp is a change
    assert not c.changed("f", a, b2)
    assert c.changed("f", a, np.ones(6))
    assert c.changed("f", np.ones(6))                 # fewer arrays
    assert c.changed("g", a, b)                       # another key has its own history


def test_the_cache_keeps_its_own_copy():
    c = _UploadCache()
    a = np.zeros(4)
    c.changed("f", a)
    a[0] = 2.0                                        # the caller reuses its buffer
    assert c.changed("f", a)
    assert not c.changed("f", a)


def test_none_entries():
    c = _UploadCache()
    assert c.changed("f", None, np.ones(3))
    assert not c.changed("f", None, np.ones(3))
    assert c.changed("f", np.zeros(3), np.ones(3))
