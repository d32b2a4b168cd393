"""A small on-disk dataset in the reference's formats (tests only): PCM16 WAVs under mix/ s1/ s2/ and
``np.savez_compressed(embedding=(512, Tv) f32)`` lip embeddings (make_embeddings.py:69), plus the index entries
BaseDataset.__getitem__ consumes (base_dataset.py:70-98).  The writer lives beside the loaders (speech_separation_amd/io.py):
bench.py's end-to-end leg uses the same one."""
from speech_separation_amd.io import write_synthetic_dataset as make_dataset  # noqa: F401
from speech_separation_amd.io import write_wav  # noqa: F401
