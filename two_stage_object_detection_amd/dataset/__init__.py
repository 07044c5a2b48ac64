"""Mirror of the reference's ``dataset`` package for the inference input step (``dataset/transform.py``)."""
