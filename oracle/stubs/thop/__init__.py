def profile(model, inputs=(), verbose=False, **kw):
    return 0.0, 0.0
