class AttrDict(dict):
    """dict whose keyword items are also attributes (reference utils/attr.py:4-8)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__.update(kwargs)
