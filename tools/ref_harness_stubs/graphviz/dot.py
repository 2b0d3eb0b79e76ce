import contextlib


class _Dot:
    def __init__(self, name=None, **kw):
        self.name = name
        self.body = []

    def attr(self, *a, **kw):
        self.body.append(("attr", a, kw))

    def node(self, name, *a, **kw):
        self.body.append(("node", name, kw))

    def edge(self, a, b, *r, **kw):
        self.body.append(("edge", a, b, kw))

    def subgraph(self, graph=None, **kw):
        if graph is not None:
            self.body.append(("subgraph", graph))
            return None
        sub = type(self)()
        self.body.append(("subgraph", sub))
        return contextlib.nullcontext(sub)

    def render(self, *a, **kw):     # draws nothing
        return None

    def pipe(self, *a, **kw):
        return b""


class Digraph(_Dot):
    pass


class Graph(_Dot):
    pass


class Source(_Dot):
    def __init__(self, source="", **kw):
        super().__init__()
        self.source = source
