"""LinOp base (include/ginkgo/core/base/lin_op.hpp:129-160): apply(b, x) and
apply(alpha, b, beta, x) with Ginkgo's dimension validation."""
from ._lib import DimensionMismatch


class LinOp:
    def __init__(self, exec_, size):
        self.exec = exec_
        self.size = tuple(int(s) for s in size)

    def get_size(self):
        return self.size

    def get_executor(self):
        return self.exec

    def _validate(self, b, x):
        if self.size[1] != b.size[0]:
            raise DimensionMismatch(
                f"apply: operator is {self.size}, b is {b.size}")
        if self.size[0] != x.size[0]:
            raise DimensionMismatch(
                f"apply: operator is {self.size}, x is {x.size}")
        if b.size[1] != x.size[1]:
            raise DimensionMismatch(f"apply: b is {b.size}, x is {x.size}")

    def apply(self, *args):
        if len(args) == 2:
            b, x = args
            self._validate(b, x)
            self.apply_impl(b, x)
        elif len(args) == 4:
            alpha, b, beta, x = args
            self._validate(b, x)
            if alpha.size != (1, 1) or beta.size != (1, 1):
                raise DimensionMismatch("alpha and beta must be 1 x 1")
            self.apply_advanced_impl(alpha, b, beta, x)
        else:
            raise TypeError("apply(b, x) or apply(alpha, b, beta, x)")
        return x
