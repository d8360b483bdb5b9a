"""Constructor argument helpers; behaviour of the reference's utils/tools.py:5-13."""


def pair(val):
    """Wrap anything that is not already a tuple (a list is NOT a tuple -- tools.py:5-6)."""
    if isinstance(val, tuple):
        return val
    return (val, val)


def check_sizes(image_size, patch_size):
    """AssertionError unless the image tiles exactly; returns the number of patches (tools.py:8-13)."""
    ih, iw = pair(image_size)
    ph, pw = pair(patch_size)
    assert (ih % ph) == 0 and (iw % pw) == 0, 'image height and width must be divisible by patch size'
    return (ih // ph) * (iw // pw)
