import os


class PathManager:
    @staticmethod
    def mkdirs(path):
        os.makedirs(path, exist_ok=True)

    @staticmethod
    def open(path, mode="r"):
        return open(path, mode)

    @staticmethod
    def isfile(path):
        return os.path.isfile(path)

    @staticmethod
    def exists(path):
        return os.path.exists(path)

    @staticmethod
    def get_local_path(path):
        return path
