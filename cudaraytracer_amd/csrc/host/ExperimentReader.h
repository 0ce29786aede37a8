// ExperimentReader: reads the reference's experiment description files
// (cudaTracer/ExperimentReader.h:7-35, ExperimentReader.cpp:8-158) so that the
// same JSON drives this library's headless driver:
//
//   { "name": "...",
//     "phases": [ { "name": "...",
//                   "options": { "model": { "folder": "...", "fname": "...", "scale": 0.02,
//                                           "translate": [x, y, z], "transformationMatrix": [16 numbers] },
//                                "lightCnt": 4, "fov": 60.0, "width": 1920, "height": 1080, "shadowBias": 0.0001,
//                                "blockSize": 8, "maxDepth": 2, "samples": 1, "gammaCorrection": true } }, ... ] }
//
// Every phase starts from the defaults handed to the constructor and overrides
// only the keys it names.  Behaviour kept from the reference:
//  * a key of the wrong JSON type is reported and its value is still taken when
//    it can be read as the wanted type (the reference pops up a message box and
//    carries on, ExperimentReader.cpp:152-157); here the messages are collected
//    in `errors`, in the reference's wording;
//  * "fov" / "shadowBias" / "scale" must be written with a fraction or exponent
//    (rapidjson's IsFloat is false for an integer literal);
//  * "scale" then "translate" are composed in that order into the model
//    transform and replace whatever "transformationMatrix" set
//    (ExperimentReader.cpp:78-100);
//  * device.scale and device.imageAspectRatio are NOT re-derived from a phase's
//    fov / width / height: the reference keeps the values of the defaults
//    (main.cpp:59-61 runs before the reader, nothing after it).  Call
//    setView(phase.opts, w, h, fov) to re-derive them.
// Differences: at most 5 phases are kept (the reference's fixed array,
// ExperimentReader.h:14, is overrun by a sixth) and the rest reported;
// "transformationMatrix" is read as 16 row-major numbers (the reference's loop,
// ExperimentReader.cpp:71-76, writes entry i across the whole of row i and runs
// past row 3); a file that is not valid JSON is reported instead of crashing.
//
// The JSON parser below is self-contained (RFC 8259 values, \uXXXX escapes,
// integer vs. floating literals told apart the way rapidjson does).
#ifndef RT_HOST_EXPERIMENTREADER_H
#define RT_HOST_EXPERIMENTREADER_H

#include <cerrno>
#include <climits>
#include <cstdlib>
#include <fstream>
#include <ostream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "Mat.h"
#include "Options.h"

namespace rt {
namespace json {

struct Value {
    enum Type { Null, Bool, Int, Float, String, Array, Object };
    Type type = Null;
    bool boolean = false;
    double number = 0;
    std::string text;
    std::vector<Value> items;                             // Array
    std::vector<std::pair<std::string, Value>> members;  // Object, in file order

    bool isString() const { return type == String; }
    bool isArray() const { return type == Array; }
    bool isObject() const { return type == Object; }
    bool isBool() const { return type == Bool; }
    bool isInt() const { return type == Int; }      // integer literal that fits an int
    bool isFloat() const { return type == Float; }  // literal with a fraction or exponent
    bool isNumber() const { return type == Int || type == Float; }
    size_t size() const { return items.size(); }

    bool hasMember(const std::string& key) const { return find(key) != nullptr; }
    // Missing members and out-of-range items read as a Null value.
    const Value& operator[](const std::string& key) const {
        const Value* v = find(key);
        return v ? *v : nothing();
    }
    const Value& operator[](size_t i) const { return i < items.size() ? items[i] : nothing(); }

    std::string getString() const { return type == String ? text : std::string(); }
    float getFloat() const { return isNumber() ? (float)number : 0.0f; }
    int getInt() const { return isNumber() ? (int)number : 0; }
    bool getBool() const { return type == Bool && boolean; }

   private:
    const Value* find(const std::string& key) const {
        if (type != Object) return nullptr;
        for (const auto& m : members)
            if (m.first == key) return &m.second;
        return nullptr;
    }
    static const Value& nothing() {
        static const Value none;
        return none;
    }
};

class Parser {
   public:
    // Parses one JSON text.  On failure returns false and sets `error`
    // ("<what> at offset <n>"); `out` is then Null.
    static bool parse(const std::string& src, Value& out, std::string& error) {
        Parser p(src);
        Value v;
        p.skipSpace();
        if (p.value(v, 0)) {
            p.skipSpace();
            if (p.pos != src.size()) p.fail("text after the document");
        }
        if (!p.error.empty()) {
            error = p.error;
            out = Value();
            return false;
        }
        out = std::move(v);
        return true;
    }

   private:
    const std::string& s;
    size_t pos = 0;
    std::string error;
    static const int kMaxDepth = 64;

    explicit Parser(const std::string& src) : s(src) {}

    bool fail(const char* what) {
        if (error.empty()) {
            std::ostringstream m;
            m << what << " at offset " << pos;
            error = m.str();
        }
        return false;
    }
    void skipSpace() {
        while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\n' || s[pos] == '\r')) ++pos;
    }
    bool literal(const char* word) {
        size_t n = 0;
        while (word[n]) ++n;
        if (s.compare(pos, n, word) != 0) return fail("unknown literal");
        pos += n;
        return true;
    }

    bool value(Value& v, int depth) {
        if (depth > kMaxDepth) return fail("nesting too deep");
        if (pos >= s.size()) return fail("unexpected end");
        const char c = s[pos];
        if (c == '{') return object(v, depth);
        if (c == '[') return array(v, depth);
        if (c == '"') {
            v.type = Value::String;
            return string(v.text);
        }
        if (c == 't' || c == 'f') {
            v.type = Value::Bool;
            v.boolean = c == 't';
            return literal(c == 't' ? "true" : "false");
        }
        if (c == 'n') {
            v.type = Value::Null;
            return literal("null");
        }
        if (c == '-' || (c >= '0' && c <= '9')) return number(v);
        return fail("unexpected character");
    }

    bool object(Value& v, int depth) {
        v.type = Value::Object;
        ++pos;  // {
        skipSpace();
        if (pos < s.size() && s[pos] == '}') {
            ++pos;
            return true;
        }
        for (;;) {
            skipSpace();
            if (pos >= s.size() || s[pos] != '"') return fail("member name expected");
            std::string key;
            if (!string(key)) return false;
            skipSpace();
            if (pos >= s.size() || s[pos] != ':') return fail("':' expected");
            ++pos;
            skipSpace();
            Value child;
            if (!value(child, depth + 1)) return false;
            v.members.emplace_back(std::move(key), std::move(child));
            skipSpace();
            if (pos < s.size() && s[pos] == ',') {
                ++pos;
                continue;
            }
            if (pos < s.size() && s[pos] == '}') {
                ++pos;
                return true;
            }
            return fail("',' or '}' expected");
        }
    }

    bool array(Value& v, int depth) {
        v.type = Value::Array;
        ++pos;  // [
        skipSpace();
        if (pos < s.size() && s[pos] == ']') {
            ++pos;
            return true;
        }
        for (;;) {
            skipSpace();
            Value child;
            if (!value(child, depth + 1)) return false;
            v.items.push_back(std::move(child));
            skipSpace();
            if (pos < s.size() && s[pos] == ',') {
                ++pos;
                continue;
            }
            if (pos < s.size() && s[pos] == ']') {
                ++pos;
                return true;
            }
            return fail("',' or ']' expected");
        }
    }

    bool hex4(unsigned& out) {
        if (pos + 4 > s.size()) return fail("short \\u escape");
        out = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = s[pos++];
            out <<= 4;
            if (c >= '0' && c <= '9')
                out |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f')
                out |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F')
                out |= (unsigned)(c - 'A' + 10);
            else
                return fail("bad \\u escape");
        }
        return true;
    }
    static void utf8(unsigned cp, std::string& out) {
        if (cp < 0x80) {
            out += (char)cp;
        } else if (cp < 0x800) {
            out += (char)(0xC0 | (cp >> 6));
            out += (char)(0x80 | (cp & 0x3F));
        } else if (cp < 0x10000) {
            out += (char)(0xE0 | (cp >> 12));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        } else {
            out += (char)(0xF0 | (cp >> 18));
            out += (char)(0x80 | ((cp >> 12) & 0x3F));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        }
    }

    bool string(std::string& out) {
        ++pos;  // opening quote
        for (;;) {
            if (pos >= s.size()) return fail("unterminated string");
            const unsigned char c = (unsigned char)s[pos++];
            if (c == '"') return true;
            if (c < 0x20) return fail("control character in string");
            if (c != '\\') {
                out += (char)c;
                continue;
            }
            if (pos >= s.size()) return fail("unterminated escape");
            const char e = s[pos++];
            switch (e) {
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                case '/': out += '/'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'n': out += '\n'; break;
                case 'r': out += '\r'; break;
                case 't': out += '\t'; break;
                case 'u': {
                    unsigned cp = 0;
                    if (!hex4(cp)) return false;
                    if (cp >= 0xD800 && cp <= 0xDBFF) {  // high surrogate: a low one must follow
                        unsigned lo = 0;
                        if (pos + 2 > s.size() || s[pos] != '\\' || s[pos + 1] != 'u') return fail("lone surrogate");
                        pos += 2;
                        if (!hex4(lo)) return false;
                        if (lo < 0xDC00 || lo > 0xDFFF) return fail("lone surrogate");
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                        return fail("lone surrogate");
                    }
                    utf8(cp, out);
                    break;
                }
                default: return fail("unknown escape");
            }
        }
    }

    bool number(Value& v) {
        const size_t start = pos;
        bool integral = true;
        if (s[pos] == '-') ++pos;
        if (pos >= s.size()) return fail("digit expected");
        if (s[pos] == '0') {
            ++pos;
        } else if (s[pos] >= '1' && s[pos] <= '9') {
            while (pos < s.size() && s[pos] >= '0' && s[pos] <= '9') ++pos;
        } else {
            return fail("digit expected");
        }
        if (pos < s.size() && s[pos] == '.') {
            integral = false;
            ++pos;
            if (pos >= s.size() || s[pos] < '0' || s[pos] > '9') return fail("digit expected after '.'");
            while (pos < s.size() && s[pos] >= '0' && s[pos] <= '9') ++pos;
        }
        if (pos < s.size() && (s[pos] == 'e' || s[pos] == 'E')) {
            integral = false;
            ++pos;
            if (pos < s.size() && (s[pos] == '+' || s[pos] == '-')) ++pos;
            if (pos >= s.size() || s[pos] < '0' || s[pos] > '9') return fail("digit expected in exponent");
            while (pos < s.size() && s[pos] >= '0' && s[pos] <= '9') ++pos;
        }
        const std::string lit = s.substr(start, pos - start);
        v.number = std::strtod(lit.c_str(), nullptr);
        // An integer literal outside the int range is neither IsInt nor IsFloat for
        // rapidjson; it is filed under Float here so that its value stays readable.
        v.type = (integral && v.number >= (double)INT_MIN && v.number <= (double)INT_MAX) ? Value::Int : Value::Float;
        return true;
    }
};

}  // namespace json

struct ExperimentPhase {
    std::string phaseName;
    Options opts;
};

struct ExperimentParameters {
    static const int MAX_PHASES = 5;
    std::string experimentName;
    ExperimentPhase phases[MAX_PHASES];
    int phaseCnt = 0;  // phases the file filled; the others hold default-constructed options
};

class ExperimentReader {
   public:
    std::string jsonFname;
    Options defaults;
    ExperimentParameters experiment;
    std::vector<std::string> errors;  // one entry per failed check, in file order

    ExperimentReader(const std::string& jsonFname, Options defaults) : jsonFname(jsonFname), defaults(defaults) {
        std::string text;
        if (!readFile(jsonFname, text)) {
            asserts(false, "The file cannot be opened!");
            return;
        }
        parseText(text);
    }

    // Parses `jsonText` directly; `label` stands in for the file name in messages.
    static ExperimentReader fromString(const std::string& jsonText, Options defaults, const std::string& label = "<string>") {
        ExperimentReader r(defaults, label);
        r.parseText(jsonText);
        return r;
    }

    bool ok() const { return errors.empty(); }

    // Whole file into a string, line ends normalised to '\n' (ExperimentReader.cpp:16-28).
    static bool readFile(const std::string& fname, std::string& out) {
        std::ifstream f(fname);
        if (!f.is_open()) return false;
        std::string line;
        while (std::getline(f, line)) {
            out += line;
            out += '\n';
        }
        return true;
    }

   private:
    ExperimentReader(Options defaults, const std::string& label) : jsonFname(label), defaults(defaults) {}

    void parseText(const std::string& text) {
        json::Value doc;
        std::string syntax;
        if (!json::Parser::parse(text, doc, syntax)) {
            asserts(false, "Not a JSON document: " + syntax + ".");
            return;
        }
        parseJson(doc);
    }

    // ExperimentReader.cpp:30-45
    void parseJson(const json::Value& doc) {
        ExperimentParameters exp;
        asserts(doc["name"].isString(), "The experiment must have a name, of type string!");
        exp.experimentName = doc["name"].getString();
        asserts(doc["phases"].isArray(), "The experiment must have a list of phases!");
        const json::Value& phases = doc["phases"];
        asserts(phases.size() <= (size_t)ExperimentParameters::MAX_PHASES, "The experiment phase list must not be longer than 5!");
        for (size_t i = 0; i < phases.size() && i < (size_t)ExperimentParameters::MAX_PHASES; ++i) {
            exp.phases[i] = parsePhase(phases[i]);
            exp.phaseCnt = (int)i + 1;
        }
        experiment = exp;
    }

    // ExperimentReader.cpp:47-56
    ExperimentPhase parsePhase(const json::Value& phaseJson) {
        ExperimentPhase phase;
        asserts(phaseJson["name"].isString(), "The phase must have a name, of type string!");
        phase.phaseName = phaseJson["name"].getString();
        phase.opts = defaults;
        parseOptions(phaseJson["options"], phase.opts);
        return phase;
    }

    // ExperimentReader.cpp:58-111
    void parseOptions(const json::Value& optionsJson, Options& opts) {
        if (optionsJson.hasMember("model") && optionsJson["model"].isObject()) {
            const json::Value& model = optionsJson["model"];
            opts.host.model.folder = model["folder"].getString();
            opts.host.model.fname = model["fname"].getString();

            if (model.hasMember("transformationMatrix") && model["transformationMatrix"].isArray()) {
                const json::Value& m = model["transformationMatrix"];
                asserts(m.size() == 16, "The translate model.transformationMatrix must be of length 16!");
                for (size_t i = 0; i < m.size() && i < 16; ++i) opts.host.model.transform[(int)(i / 4)][i % 4] = m[i].getFloat();
            }
            std::vector<Mat44f> transformations;
            if (model.hasMember("scale")) {
                asserts(model["scale"].isFloat(), "The translate model.member must be a float!");
                transformations.push_back(Mat44f::scale(model["scale"].getFloat()));
            }
            if (model.hasMember("translate")) {
                asserts(model["translate"].isArray(), "The translate model.member must be an array!");
                asserts(model["translate"].size() == 3, "The translate model->translate must be of length 3!");
                transformations.push_back(
                    Mat44f::translate(model["translate"][0].getFloat(), model["translate"][1].getFloat(), model["translate"][2].getFloat()));
            }
            if (!transformations.empty()) {
                Mat44f finalMat = Mat44f::identity();
                for (const Mat44f& t : transformations) finalMat = finalMat * t;
                opts.host.model.transform = finalMat;
            }
        }
        parseInt(optionsJson, "lightCnt", opts.host.lightCnt);
        parseFloat(optionsJson, "fov", opts.host.fov);
        parseInt(optionsJson, "width", opts.device.width);
        parseInt(optionsJson, "height", opts.device.height);
        parseFloat(optionsJson, "shadowBias", opts.device.shadowBias);
        parseInt(optionsJson, "blockSize", opts.device.blockSize);
        parseInt(optionsJson, "maxDepth", opts.device.maxDepth);
        parseInt(optionsJson, "samples", opts.device.samples);
        parseBool(optionsJson, "gammaCorrection", opts.device.gammaCorrection);
    }

    // ExperimentReader.cpp:113-150.  A value of another type is reported; a number is
    // still converted (GetInt on 2.5 would abort inside rapidjson; here it truncates).
    void parseInt(const json::Value& element, const std::string& key, int& out_target, bool mandatory = false) {
        if (mandatory) asserts(element.hasMember(key), "'" + key + "' must exist!");
        if (element.hasMember(key)) {
            asserts(element[key].isInt(), "'" + key + "' must be of type int!");
            if (element[key].isNumber()) out_target = element[key].getInt();
        }
    }
    void parseFloat(const json::Value& element, const std::string& key, float& out_target, bool mandatory = false) {
        if (mandatory) asserts(element.hasMember(key), "'" + key + "' must exist!");
        if (element.hasMember(key)) {
            asserts(element[key].isFloat(), "'" + key + "' must be of type float!");
            if (element[key].isNumber()) out_target = element[key].getFloat();
        }
    }
    void parseBool(const json::Value& element, const std::string& key, bool& out_target, bool mandatory = false) {
        if (mandatory) asserts(element.hasMember(key), "'" + key + "' must exist!");
        if (element.hasMember(key)) {
            asserts(element[key].isBool(), "'" + key + "' must be of type bool!");
            if (element[key].isBool()) out_target = element[key].getBool();
        }
    }

    // ExperimentReader.cpp:152-158: report and carry on.
    void asserts(bool assertion, const std::string& text) {
        if (!assertion) errors.push_back("Failed to parse json file: '" + jsonFname + "'.\n" + text);
    }
};

// ExperimentReader.cpp:160-170, for the phases the file filled.
inline std::ostream& operator<<(std::ostream& os, const ExperimentReader& r) {
    os << "file: " << r.jsonFname << std::endl;
    os << "exp name: " << r.experiment.experimentName << std::endl;
    for (int i = 0; i < r.experiment.phaseCnt; ++i) {
        os << "\tphase name: " << r.experiment.phases[i].phaseName << std::endl;
        os << "\t\tphase model: " << r.experiment.phases[i].opts.host.model.fname << std::endl;
    }
    return os;
}

}  // namespace rt
#endif
