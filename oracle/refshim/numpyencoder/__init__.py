import json


class NumpyEncoder(json.JSONEncoder):
    pass
