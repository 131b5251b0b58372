from typing import Callable

FailureCallback = Callable[[], None]
