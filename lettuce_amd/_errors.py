"""Exception and warning classes (lettuce/util/utility.py:21-34)."""

__all__ = ["LettuceException", "LettuceWarning", "InefficientCodeWarning", "ExperimentalWarning"]


class LettuceException(Exception):
    pass


class LettuceWarning(UserWarning):
    pass


class InefficientCodeWarning(LettuceWarning):
    pass


class ExperimentalWarning(LettuceWarning):
    pass
