def register(*args, **kwargs):
    """No-op: the stub has no environment registry."""
    return None
