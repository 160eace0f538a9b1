"""Oracle = test infrastructure (CPU restatement of the reference path). Not product code."""
