"""Host-side assertions.  The reference raises ``RuntimeError('Assertion error (file:line): <cond>')`` from
``DG_HOST_ASSERT`` (csrc/utils/exception.hpp:12-35); the Python host layer keeps that message shape."""
import inspect
import os


def host_assert(cond: bool, what: str) -> None:
    if not cond:
        frame = inspect.stack()[1]
        raise RuntimeError(f'Assertion error ({os.path.basename(frame.filename)}:{frame.lineno}): {what}')
