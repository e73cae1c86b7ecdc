"""CPU: the driver's build() entry point runs here (hipcc cross-compiles without a GPU) and leaves a
library whose ABI version and symbols match the bindings."""
import __graft_entry__ as entry


def test_build_entry_point():
    entry.build()
