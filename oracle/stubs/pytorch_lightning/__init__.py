__version__ = "stub-2.1.3"
