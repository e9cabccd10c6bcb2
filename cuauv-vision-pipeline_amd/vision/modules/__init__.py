"""Mirror of the reference modules/ package: only the modules whose arithmetic is on the accelerated path."""
