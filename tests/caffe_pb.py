"""Independent codec for the Caffe messages the snapshots use, built at run time with
python-protobuf from hand-written descriptors (no protoc in the image).  Field numbers are
caffe.proto's (BVLC/caffe @2ef5847); only the fields the reference's snapshots
rely on are declared, unknown fields are preserved/ignored by protobuf."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=_F.LABEL_OPTIONAL, type_name=None, packed=False):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    if packed:
        f.options.packed = True


def _build():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "caffe_min.proto"; fd.package = "caffe"; fd.syntax = "proto2"
    m = fd.message_type.add(); m.name = "BlobShape"
    _field(m, "dim", 1, _F.TYPE_INT64, _F.LABEL_REPEATED, packed=True)
    m = fd.message_type.add(); m.name = "BlobProto"
    _field(m, "shape", 7, _F.TYPE_MESSAGE, type_name=".caffe.BlobShape")
    _field(m, "data", 5, _F.TYPE_FLOAT, _F.LABEL_REPEATED, packed=True)
    _field(m, "diff", 6, _F.TYPE_FLOAT, _F.LABEL_REPEATED, packed=True)
    for i, n in enumerate(("num", "channels", "height", "width")):
        _field(m, n, i + 1, _F.TYPE_INT32)
    m = fd.message_type.add(); m.name = "FillerParameter"
    _field(m, "type", 1, _F.TYPE_STRING); _field(m, "value", 2, _F.TYPE_FLOAT); _field(m, "std", 6, _F.TYPE_FLOAT)
    m = fd.message_type.add(); m.name = "InnerProductParameter"
    _field(m, "num_output", 1, _F.TYPE_UINT32); _field(m, "bias_term", 2, _F.TYPE_BOOL)
    _field(m, "weight_filler", 3, _F.TYPE_MESSAGE, type_name=".caffe.FillerParameter")
    m = fd.message_type.add(); m.name = "MemoryDataParameter"
    for i, n in enumerate(("batch_size", "channels", "height", "width")):
        _field(m, n, i + 1, _F.TYPE_UINT32)
    m = fd.message_type.add(); m.name = "ReLUParameter"
    _field(m, "negative_slope", 1, _F.TYPE_FLOAT)
    m = fd.message_type.add(); m.name = "ConcatParameter"
    _field(m, "concat_dim", 1, _F.TYPE_UINT32); _field(m, "axis", 2, _F.TYPE_INT32)
    m = fd.message_type.add(); m.name = "LayerParameter"
    _field(m, "name", 1, _F.TYPE_STRING); _field(m, "type", 2, _F.TYPE_STRING)
    _field(m, "bottom", 3, _F.TYPE_STRING, _F.LABEL_REPEATED); _field(m, "top", 4, _F.TYPE_STRING, _F.LABEL_REPEATED)
    _field(m, "blobs", 7, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, type_name=".caffe.BlobProto")
    _field(m, "phase", 10, _F.TYPE_INT32)               # enum Phase { TRAIN = 0; TEST = 1; }
    _field(m, "concat_param", 104, _F.TYPE_MESSAGE, type_name=".caffe.ConcatParameter")
    _field(m, "inner_product_param", 117, _F.TYPE_MESSAGE, type_name=".caffe.InnerProductParameter")
    _field(m, "memory_data_param", 119, _F.TYPE_MESSAGE, type_name=".caffe.MemoryDataParameter")
    _field(m, "relu_param", 123, _F.TYPE_MESSAGE, type_name=".caffe.ReLUParameter")
    m = fd.message_type.add(); m.name = "NetParameter"
    _field(m, "name", 1, _F.TYPE_STRING); _field(m, "force_backward", 5, _F.TYPE_BOOL)
    _field(m, "layer", 100, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, type_name=".caffe.LayerParameter")
    m = fd.message_type.add(); m.name = "SolverState"
    _field(m, "iter", 1, _F.TYPE_INT32); _field(m, "learned_net", 2, _F.TYPE_STRING)
    _field(m, "history", 3, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, type_name=".caffe.BlobProto")
    _field(m, "current_step", 4, _F.TYPE_INT32)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("caffe." + n))
    return {n: get(n) for n in ("BlobShape", "BlobProto", "LayerParameter", "NetParameter", "SolverState")}


MSG = _build()
NetParameter, SolverState, LayerParameter, BlobProto = (MSG[k] for k in ("NetParameter", "SolverState", "LayerParameter", "BlobProto"))
