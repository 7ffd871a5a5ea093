"""Integer helpers shared by the host modules (kept dependency-free to avoid import cycles)."""


def ceil_div(x: int, y: int) -> int:
    return -(-x // y)


def align(x: int, y: int) -> int:
    return ceil_div(x, y) * y
