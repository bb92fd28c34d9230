"""Call-signature records of a module's public surface (golden G20: the reference's; tests: the nerf_amd mirror's)."""
import inspect

MODULES = ("nerf_base", "nerf_helper", "mip_methods", "mip_model", "procedures", "utils", "addtional", "ref_model", "ref_func", "dataset",
           "param_com", "local_shuffler", "timer")
ENTRY_SCRIPTS = ("train.py", "ddp_train.py", "model_average.py")


def entry_imports(path):
    """{module: [names]} of every `from nerf.<module> import ...` statement of a script (golden G23: the reference's entry scripts;
    `*` is recorded as "*")."""
    import ast
    with open(path) as f:
        tree = ast.parse(f.read())
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("nerf."):
            out.setdefault(node.module[5:], [])
            out[node.module[5:]] += [a.name for a in node.names]
    return {k: sorted(set(v)) for k, v in out.items()}


def _default(v):
    if v is inspect.Parameter.empty:
        return None
    if v is None or isinstance(v, (bool, int, float, str)):
        return ["value", repr(v)]
    if isinstance(v, (tuple, list)):
        return ["value", repr(tuple(v))]
    return ["object", getattr(v, "__name__", type(v).__name__)]      # e.g. F.relu -> "relu"


def signature_record(fn):
    try:
        sig = inspect.signature(fn)
    except (TypeError, ValueError):
        return None
    return [[p.name, p.kind.name, _default(p.default)] for p in sig.parameters.values()]


def module_signatures(mod, modname):
    """{name: params} for functions, {Class.method: params} for the methods a class itself defines (incl. __init__ / forward / staticmethods)"""
    out = {}
    for name, obj in vars(mod).items():
        if name.startswith("_") or getattr(obj, "__module__", None) != mod.__name__:
            continue
        if inspect.isfunction(obj):
            rec = signature_record(obj)
            if rec is not None:
                out[name] = rec
        elif inspect.isclass(obj):
            out[name] = "class"
            for mname, m in vars(obj).items():
                if mname.startswith("_") and mname != "__init__":
                    continue
                f = m.__func__ if isinstance(m, (staticmethod, classmethod)) else m
                if inspect.isfunction(f):
                    rec = signature_record(f)
                    if rec is not None:
                        out["%s.%s" % (name, mname)] = rec
    return out
