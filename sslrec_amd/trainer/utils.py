"""Small helpers of the training loop (behaviour of the reference's trainer/utils.py:1-19)."""
import functools
import logging


class DisabledSummaryWriter:
    """Absorbs any TensorBoard-writer call: every attribute and every call returns the object itself."""

    def __init__(self, *_, **__):
        pass

    def __getattr__(self, _name):
        return self

    def __call__(self, *_, **__):
        return self


def log_exceptions(method):
    """Decorator for Trainer entry points: write the traceback to the run log, then re-raise."""
    @functools.wraps(method)
    def guarded(*args, **kwargs):
        try:
            return method(*args, **kwargs)
        except Exception:
            logging.getLogger('train_logger').exception('training aborted')
            raise
    return guarded
