"""TEST INFRASTRUCTURE.  The synthetic training set in the reference's on-disk layout used by tests/test_dataset_reader.py and
oracle/make_feed_goldens.py; the writer itself lives with the other synthetic-input generators (giga_amd/synth.py)."""
from giga_amd.synth import write_training_set as write_dataset  # noqa: F401
