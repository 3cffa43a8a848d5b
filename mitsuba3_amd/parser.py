"""Scene description front end: Mitsuba XML -> the nested dict `load_dict` consumes (src/core/parser.cpp:280-1370, the XML half of
`mi.parser`; `mi.load_file` / `mi.load_string`, src/core/python/parser.cpp).  Host-side only, outside the hot path; written against
`xml.parsers.expat` (the reference uses pugixml, `ext/pugixml` is an empty submodule here).

Supported, with the reference's semantics: object tags (`scene`, `shape`, `bsdf`, `emitter`, `sensor`, `film`, `sampler`, `rfilter`,
`integrator`, `texture`, ...) with `type` / `id` / `name`; property tags `float integer boolean string vector point rgb spectrum
transform` (+ `translate rotate scale lookat matrix`, composed left to right = each one multiplies from the left, parser.cpp:1040-1130);
`ref`, `default` + `$name` substitution (keyword arguments override defaults, `\\$` escapes), `include` (a `<scene>` root merges its
children), `alias`, `path`; the camelCase -> snake_case upgrade of pre-2.0 scenes; line / column in error messages."""
import os
import re
import xml.parsers.expat as expat

import numpy as np

from . import core

OBJECT_TAGS = {"scene", "shape", "bsdf", "emitter", "sensor", "film", "sampler", "rfilter", "integrator", "texture", "medium", "phase", "volume"}
PROPERTY_TAGS = {"float", "integer", "boolean", "string", "vector", "point", "rgb", "spectrum", "transform", "ref", "default", "include", "alias", "path"}
TRANSFORM_OPS = {"translate", "rotate", "scale", "lookat", "matrix"}
EXPECTED_ATTRS = {
    "float": {"name", "value"}, "integer": {"name", "value"}, "boolean": {"name", "value"}, "string": {"name", "value"},
    "vector": {"name", "value", "x", "y", "z"}, "point": {"name", "value", "x", "y", "z"}, "rgb": {"name", "value"},
    "spectrum": {"name", "value", "filename", "type", "id"}, "transform": {"name"}, "ref": {"name", "id"}, "default": {"name", "value"},
    "include": {"filename"}, "alias": {"id", "as"}, "path": {"value"},
    "translate": {"value", "x", "y", "z"}, "scale": {"value", "x", "y", "z"}, "rotate": {"value", "x", "y", "z", "angle"},
    "lookat": {"origin", "target", "up"}, "matrix": {"value"},
}


class ParserConfig:
    """mi.parser.ParserConfig: `unused_parameters` = 'error' | 'warn' | 'debug' (parser.h:60-100)"""
    def __init__(self, variant="hip_ad_rgb"):
        self.variant = variant
        self.unused_parameters = "error"
        self.unused_properties = "error"


class _Elem:
    __slots__ = ("tag", "attrs", "children", "line", "col", "src")

    def __init__(self, tag, attrs, line, col, src):
        self.tag, self.attrs, self.children, self.line, self.col, self.src = tag, attrs, [], line, col, src


class _Error(RuntimeError):
    pass


def _fail(e, msg):
    where = "string" if e.src is None else 'file "%s"' % e.src
    raise _Error("Error while loading %s (line %d, col %d): %s" % (where, e.line, e.col, msg))


def _parse_tree(text, src):
    p = expat.ParserCreate()
    stack, root = [], []

    def start(tag, attrs):
        e = _Elem(tag, attrs, p.CurrentLineNumber, p.CurrentColumnNumber + 2, src)     # column of the tag name, like pugixml offsets
        (stack[-1].children if stack else root).append(e); stack.append(e)

    def end(tag):
        stack.pop()
    p.StartElementHandler, p.EndElementHandler = start, end
    try:
        p.Parse(text, True)
    except expat.ExpatError as ex:
        msg = expat.errors.messages[ex.code] if ex.code < len(expat.errors.messages) else str(ex)
        if "no element found" in msg:
            raise _Error("XML parsing failed: No document element found")
        raise _Error("XML parsing failed: %s (line %d, col %d)" % (msg, ex.lineno, ex.offset + 1))
    if not root:
        raise _Error("XML parsing failed: No document element found")
    return root[0]


def _floats(e, s, what="floating point"):
    out = []
    for tok in re.split(r"[\s,]+", s.strip()):
        if tok == "":
            continue
        try:
            out.append(float(tok))
        except ValueError:
            _fail(e, 'could not parse %s value "%s"' % (what, tok))
    return out


def _camel_to_snake(name):
    return re.sub(r"([a-z0-9])([A-Z])", lambda m: m.group(1) + "_" + m.group(2).lower(), name)


class _Parser:
    def __init__(self, config, params, search_paths):
        self.config, self.params, self.used = config, dict(params), set()
        self.ids = {}                 # id -> element (duplicate detection)
        self.aliases = {}
        self.paths = list(search_paths)
        self.depth = 0
        self.version = (3, 0, 0)

    # -- $substitution (parser.cpp:430-500): longest parameter names first, `\$` is a literal dollar
    def subst(self, e, s):
        if "$" not in s:
            return s
        out, i = "", 0
        while i < len(s):
            if s[i] == "\\" and i + 1 < len(s) and s[i + 1] == "$":
                out += "$"; i += 2; continue
            if s[i] != "$":
                out += s[i]; i += 1; continue
            best = None
            for k in sorted(self.params, key=lambda k: (-len(k), k)):
                if s.startswith(k, i + 1):
                    best = k; break
            if best is None:
                m = re.match(r"\$[A-Za-z0-9_]*", s[i:])
                _fail(e, "undefined parameter: %s" % m.group(0))
            self.used.add(best); out += str(self.params[best]); i += 1 + len(best)
        return out

    def attrs(self, e):
        a = {k: self.subst(e, v) for k, v in e.attrs.items()}
        exp = EXPECTED_ATTRS.get(e.tag)
        if exp is not None:
            for k in a:
                if k not in exp:
                    _fail(e, 'unexpected attribute "%s" in <%s>' % (k, e.tag))
        return a

    def resolve_file(self, e, name):
        if os.path.isabs(name) and os.path.exists(name):
            return name
        for base in self.paths:
            cand = os.path.join(base, name)
            if os.path.exists(cand):
                return cand
        _fail(e, 'file "%s" not found' % name)

    # -- values
    def vec3(self, e, a, default=0.0, allow_scalar=False):
        if "value" in a:
            if any(k in a for k in "xyz"):
                _fail(e, 'Cannot mix "value" and "x"/"y"/"z" attributes')
            v = _floats(e, a["value"])
            if allow_scalar and len(v) == 1:
                v = v * 3
            if len(v) != 3:
                _fail(e, "expected 3 values, got %d" % len(v))
            return v
        return [(_floats(e, a[k])[0] if k in a else default) for k in "xyz"]

    def transform(self, e):
        T = core.ScalarTransform4f()
        for c in e.children:
            if c.tag not in TRANSFORM_OPS:
                _fail(c, "unexpected <%s> element inside <transform>" % c.tag)
            a = self.attrs(c)
            if c.tag == "translate":
                op = core.ScalarTransform4f().translate(self.vec3(c, a))
            elif c.tag == "scale":
                op = core.ScalarTransform4f().scale(self.vec3(c, a, default=1.0, allow_scalar=True))
            elif c.tag == "rotate":
                if "angle" not in a:
                    _fail(c, 'missing attribute "angle" in <rotate>')
                op = core.ScalarTransform4f().rotate(self.vec3(c, a), _floats(c, a["angle"])[0])
            elif c.tag == "lookat":
                for k in ("origin", "target", "up"):
                    if k not in a:
                        _fail(c, 'missing attribute "%s" in <lookat>' % k)
                o, t, u = (_floats(c, a[k]) for k in ("origin", "target", "up"))
                if not (len(o) == len(t) == len(u) == 3):
                    _fail(c, "<lookat>: origin, target and up need 3 values each")
                op = core.ScalarTransform4f().look_at(o, t, u)
            else:
                v = _floats(c, a.get("value", ""))
                if len(v) == 9:
                    m = np.eye(4); m[:3, :3] = np.asarray(v).reshape(3, 3)
                elif len(v) == 16:
                    m = np.asarray(v).reshape(4, 4)
                else:
                    _fail(c, "matrix must have 9 or 16 values")
                inv_t = np.linalg.inv(m).T
                op = core.ScalarTransform4f(np.concatenate([m.ravel(), inv_t.ravel()]).astype(np.float32))
            T = op @ T                                    # every new operation multiplies from the left
        return T

    def prop_name(self, e, a):
        if "name" not in a:
            _fail(e, 'missing attribute "name" in <%s>' % e.tag)
        n = a["name"]
        return _camel_to_snake(n) if self.version < (2, 0, 0) else n

    def put(self, e, props, name, value):
        if name in props:
            _fail(e, 'Property "%s" was specified multiple times' % name)
        props[name] = value

    def property(self, e, props):
        a = self.attrs(e)
        for c in e.children:
            if e.tag != "transform":
                _fail(c, "<%s> element cannot occur as child of a property" % c.tag)
        t = e.tag
        if t == "default":
            if "name" not in a or "value" not in a:
                _fail(e, '<default> needs "name" and "value"')
            self.params.setdefault(a["name"], a["value"])
            if a["name"] not in self._kwargs:
                self.used.add(a["name"])          # a default that nobody references is not an "unused parameter"
            return
        if t == "path":
            d = a.get("value", "")
            d = d if os.path.isabs(d) else os.path.join(self.paths[0] if self.paths else ".", d)
            if not os.path.isdir(d):
                _fail(e, '<path>: folder "%s" not found' % d)
            self.paths.insert(0, d); return
        if t == "alias":
            if a.get("id") not in self.ids:
                _fail(e, 'referenced id "%s" not found' % a.get("id"))
            if a.get("as") in self.ids or a.get("as") in self.aliases:
                _fail(e, 'duplicate ID: "%s"' % a.get("as"))
            self.aliases[a["as"]] = a["id"]; return
        if t == "ref":
            if "id" not in a:
                _fail(e, 'missing attribute "id" in <ref>')
            rid = self.aliases.get(a["id"], a["id"])
            name = a.get("name", "_ref_%d" % len(props))
            self.put(e, props, name, {"type": "ref", "id": rid}); return
        name = self.prop_name(e, a)
        if t == "transform":
            self.put(e, props, name, self.transform(e)); return
        if t in ("vector", "point"):
            self.put(e, props, name, self.vec3(e, a)); return
        if "value" not in a and not (t == "spectrum" and "filename" in a):
            _fail(e, 'missing attribute "value" in <%s>' % t)
        v = a.get("value", "")
        if t == "float":
            f = _floats(e, v)
            if len(f) != 1:
                _fail(e, 'could not parse floating point value "%s"' % v)
            self.put(e, props, name, f[0])
        elif t == "integer":
            try:
                self.put(e, props, name, int(v.strip()))
            except ValueError:
                _fail(e, 'could not parse integer value "%s"' % v)
        elif t == "boolean":
            if v not in ("true", "false"):
                _fail(e, 'could not parse boolean value "%s" -- must be "true" or "false"' % v)
            self.put(e, props, name, v == "true")
        elif t == "string":
            self.put(e, props, name, v)
        elif t == "rgb":
            f = _floats(e, v)
            if len(f) == 1:
                f = f * 3
            if len(f) != 3:
                _fail(e, "'rgb' tag requires one or three values (got \"%s\")" % v)
            self.put(e, props, name, {"type": "rgb", "value": f})
        elif t == "spectrum":
            if "filename" in a or ":" in v:
                _fail(e, "wavelength-dependent <spectrum> data is not supported by the hip_ad_rgb variant (rgb only)")
            f = _floats(e, v)
            if len(f) != 1:
                _fail(e, "'spectrum' tag requires one value or wavelength:value pairs")
            self.put(e, props, name, {"type": "rgb", "value": f * 3})      # a uniform spectrum is the grey colour in RGB variants

    def include(self, e, props):
        a = self.attrs(e)
        if "filename" not in a:
            _fail(e, 'missing attribute "filename" in <include>')
        path = self.resolve_file(e, a["filename"])
        if self.depth > 15:
            _fail(e, "exceeded maximum include recursion depth of 15")
        with open(path, "r") as f:
            text = f.read()
        try:
            root = _parse_tree(text, path)
        except _Error as ex:
            _fail(e, 'while processing <include>: ' + str(ex))
        self.paths.insert(0, os.path.dirname(os.path.abspath(path))); self.depth += 1
        try:
            if root.tag == "scene":                   # merge the children of an included scene into the parent
                self.check_version(root)
                for c in root.children:
                    self.child(c, props)
            else:
                self.child(root, props, top=True)
        finally:
            self.paths.pop(0); self.depth -= 1

    def check_version(self, e):
        v = e.attrs.get("version")
        if v is None:
            _fail(e, 'missing attribute "version" in <%s>' % e.tag)
        v = self.subst(e, v)
        if not re.fullmatch(r"\d+\.\d+\.\d+", v):
            _fail(e, 'Invalid version number "%s"' % v)
        self.version = tuple(int(x) for x in v.split("."))

    def obj(self, e, top=False):
        a = {k: self.subst(e, v) for k, v in e.attrs.items()}
        for k in a:
            if k not in ("type", "id", "name", "version"):
                _fail(e, 'unexpected attribute "%s" in <%s>' % (k, e.tag))
        if e.tag != "scene" and "type" not in a:
            _fail(e, 'missing attribute "type" in <%s>' % e.tag)
        props = {"type": a.get("type", "scene")}
        if "id" in a:
            if a["id"] in self.ids:
                prev = self.ids[a["id"]]
                _fail(e, 'duplicate ID: "%s" (previous was at line %d, col %d)' % (a["id"], prev.line, prev.col))
            self.ids[a["id"]] = e
        for c in e.children:
            self.child(c, props)
        return props

    def child(self, c, props, top=False):
        if c.tag in OBJECT_TAGS:
            sub = self.obj(c)
            a = c.attrs
            key = self.subst(c, a["name"]) if "name" in a else (self.subst(c, a["id"]) if "id" in a else "_arg_%d" % sum(1 for k in props if k.startswith("_arg_")))
            if self.version < (2, 0, 0) and "name" in a:
                key = _camel_to_snake(key)
            self.put(c, props, key, sub)
        elif c.tag == "include":
            self.include(c, props)
        elif c.tag in PROPERTY_TAGS:
            self.property(c, props)
        elif c.tag in TRANSFORM_OPS:
            _fail(c, "transform operations can only occur inside a <transform> element")
        else:
            _fail(c, "encountered an unsupported XML element: <%s>" % c.tag)

    def run(self, root, kwargs):
        self._kwargs = set(kwargs)
        if root.tag not in OBJECT_TAGS:
            _fail(root, "encountered an unsupported XML element: <%s>" % root.tag)
        self.check_version(root)
        d = self.obj(root, top=True)
        unused = [k for k in kwargs if k not in self.used]
        if unused and self.config.unused_parameters == "error":
            unused.sort(key=lambda k: (-len(k), k))
            raise _Error("Found unused parameters:\n" + "\n".join("  - $%s=%s" % (k, kwargs[k]) for k in unused))
        return d


def parse_string(config, text, **kwargs):
    """mi.parser.parse_string: XML text -> nested dict (the reference returns a ParserState; the dict is its `load_dict` equivalent)"""
    config = config or ParserConfig()
    root = _parse_tree(text, None)
    return _Parser(config, {k: str(v) for k, v in kwargs.items()}, [os.getcwd()]).run(root, kwargs)


def parse_file(config, filename, **kwargs):
    config = config or ParserConfig()
    filename = os.fspath(filename)
    if not os.path.exists(filename):
        raise RuntimeError('"%s": file does not exist!' % filename)
    with open(filename, "r") as f:
        text = f.read()
    root = _parse_tree(text, filename)
    return _Parser(config, {k: str(v) for k, v in kwargs.items()}, [os.path.dirname(os.path.abspath(filename))]).run(root, kwargs)


def _resolve_filenames(d, base_dirs):
    """the FileResolver step of plugin construction: relative `filename` properties are looked up next to the scene file"""
    for k, v in list(d.items()):
        if isinstance(v, dict):
            _resolve_filenames(v, base_dirs)
        elif k == "filename" and isinstance(v, str) and not os.path.isabs(v):
            for b in base_dirs:
                if os.path.exists(os.path.join(b, v)):
                    d[k] = os.path.join(b, v); break
    return d


def load_string(text, **kwargs):
    """mi.load_string (src/core/python/parser.cpp)"""
    return core.load_dict(parse_string(None, text, **kwargs))


def load_file(filename, **kwargs):
    """mi.load_file (src/core/python/parser.cpp): XML scene description -> instantiated object"""
    d = parse_file(None, filename, **kwargs)
    return core.load_dict(_resolve_filenames(d, [os.path.dirname(os.path.abspath(os.fspath(filename)))]))
