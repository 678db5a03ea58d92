class UnifiedLogger:
    def __init__(self, *a, **k):
        pass
